#!/usr/bin/env python
"""What a GPU process leaves for the NEXT process's start-up: run variant X of tools/probes/hip_init_probe, then at once a
process that only calls hipInit(0); report X's wall / inside and the follower's hipInit (50 ms on a quiet GPU)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = "/tmp/hip_init_probe"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tools/probes/hip_init_probe.cpp"), "-ldl"])
def run(env):
    t0 = time.perf_counter()
    out = subprocess.run([exe], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    wall = 1e3 * (time.perf_counter() - t0)
    ins = [float(l.split()[1]) for l in out.splitlines() if l.startswith("total")]
    ini = [float(l.split()[1]) for l in out.splitlines() if l.startswith("hipInit(0)")]
    return wall, (ins or [0])[0], (ini or [0])[0]
V = [("init_only", {"PROBE_INIT_ONLY": "1"}),
     ("alloc0", {"PROBE_FAST_EXIT": "1", "PROBE_ALLOC_MB": "0"}),
     ("alloc600", {"PROBE_FAST_EXIT": "1"}),
     ("alloc600_touch", {"PROBE_FAST_EXIT": "1", "PROBE_TOUCH": "1"}),
     ("alloc4000_touch", {"PROBE_FAST_EXIT": "1", "PROBE_ALLOC_MB": "4000", "PROBE_TOUCH": "1"}),
     ("alloc4000_touch_free", {"PROBE_FAST_EXIT": "1", "PROBE_ALLOC_MB": "4000", "PROBE_TOUCH": "1", "PROBE_FREE": "1"}),
     ("alloc600_reset", {"PROBE_FAST_EXIT": "1", "PROBE_RESET": "1"}),
     ("alloc600_normal_exit", {}),
     ("one_queue", {"PROBE_FAST_EXIT": "1", "GPU_MAX_HW_QUEUES": "1"}),
     ("sdma_off", {"PROBE_FAST_EXIT": "1", "HSA_ENABLE_SDMA": "0"}),
     ("one_queue_sdma_off", {"PROBE_FAST_EXIT": "1", "GPU_MAX_HW_QUEUES": "1", "HSA_ENABLE_SDMA": "0"}),
     ("no_interrupt", {"PROBE_FAST_EXIT": "1", "HSA_ENABLE_INTERRUPT": "0"}),
     ("scratch_small", {"PROBE_FAST_EXIT": "1", "HSA_SCRATCH_SINGLE_LIMIT": "0", "HSA_NO_SCRATCH_RECLAIM": "1"}),
     ("cu_mask_half", {"PROBE_FAST_EXIT": "1", "HSA_CU_MASK": "0:0-127"}),
     ("sleep150_then_exit", {"PROBE_FAST_EXIT": "1", "PROBE_SLEEP_MS": "150"}),
     ("shutdown_then_150ms", {"PROBE_EARLY_SHUTDOWN": "150"}),
     ("reset_shutdown_then_150ms", {"PROBE_EARLY_SHUTDOWN": "150", "PROBE_RESET_FIRST": "1"})]
import sys
if len(sys.argv) > 1:
    V = [v for v in V if v[0] in sys.argv[1:]]
for name, env in V:
    rows = []
    for _ in range(5):
        time.sleep(0.5)
        w, i, h = run(env)
        w2, i2, h2 = run({"PROBE_INIT_ONLY": "1"})
        rows.append((w, i, h, h2, w2))
    f = lambda k: " ".join("%4.0f" % r[k] for r in rows)
    print("%-22s wall %s | inside %s | own hipInit %s | follower hipInit %s | follower wall %s" % (name, f(0), f(1), f(2), f(3), f(4)), flush=True)
