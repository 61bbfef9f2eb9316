#!/usr/bin/env python
"""HIP start-up and process teardown of a minimal process under different runtime settings (through gpurun):
wall clock of tools/probes/hip_init_probe (hipInit, first queue, one launch, a 600 MB allocation) seen by the parent, next to
the probe's own total; `gap` = what happens before main() and after exit (loader, kernel-side KFD / address-space teardown).
Each setting: 5 runs back to back, then 5 with 0.4 s pauses (no previous process's teardown under the start-up)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = "/tmp/hip_init_probe"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-o", exe, os.path.join(ROOT, "tools/probes/hip_init_probe.cpp"), "-ldl"])
SETTINGS = [("default", {}), ("fast_exit", {"PROBE_FAST_EXIT": "1"}),
            ("sdma_off", {"PROBE_FAST_EXIT": "1", "HSA_ENABLE_SDMA": "0"}),
            ("one_queue", {"PROBE_FAST_EXIT": "1", "GPU_MAX_HW_QUEUES": "1"}),
            ("rocr_visible0", {"PROBE_FAST_EXIT": "1", "ROCR_VISIBLE_DEVICES": "0"}),
            ("no_interrupt", {"PROBE_FAST_EXIT": "1", "HSA_ENABLE_INTERRUPT": "0"}),
            ("no_dm_heap", {"PROBE_FAST_EXIT": "1", "HIP_INITIAL_DM_SIZE": "0"}),
            ("all", {"PROBE_FAST_EXIT": "1", "HSA_ENABLE_SDMA": "0", "GPU_MAX_HW_QUEUES": "1", "ROCR_VISIBLE_DEVICES": "0", "HIP_INITIAL_DM_SIZE": "0"})]
for name, env in SETTINGS:
    for pause in (0.0, 0.4):
        walls, insides, inits = [], [], []
        for _ in range(5):
            time.sleep(pause)
            t0 = time.perf_counter()
            out = subprocess.run([exe], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            walls.append(1e3 * (time.perf_counter() - t0))
            for line in out.splitlines():
                if line.startswith("total"):
                    insides.append(float(line.split()[1]))
                if line.startswith("hipInit(0)"):
                    inits.append(float(line.split()[1]))
        f = lambda v: " ".join("%4.0f" % x for x in v)
        print("%-14s pause %.1f | wall %s | inside %s | hipInit %s | gap %s" % (name, pause, f(walls), f(insides), f(inits), f([w - i for w, i in zip(walls, insides)])), flush=True)
