cd /tmp; rm -rf pl && mkdir pl && cd pl && python - <<EOF
import sys, os, time, subprocess
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from hinge_amd import synth
d = synth.generate(synth.CONFIGS["cfg2_ecoli160"]); synth.write_dataset(d, ".", "G", write_bases=False)
open("nominal.ini","w").write("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")
H = os.path.join(os.environ["GRAFT_REPO_ROOT"], "hinge_amd/bin/hinge")
subprocess.run([H,"filter","--db","G","--las","G.las","-x","H","--config","nominal.ini"],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
for n in ("1","2","4","8","4","1"):
  ts=[]
  for k in range(3):
    t=time.time(); r=subprocess.run([H,"maximal","--db","G","--las","G.las","-x","H","--config","nominal.ini"],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,HINGE_HOST_TIMING="1",HINGE_UPLOAD_THREADS=n)); ts.append(round(time.time()-t,3))
  print("threads", n, ts, [l.split()[-2] for l in r.stderr.decode().split("\n") if "traces H2D" in l])
import hashlib
print(hashlib.sha256(open("H.max","rb").read()).hexdigest()[:16], hashlib.sha256(open("H.contained.txt","rb").read()).hexdigest()[:16])
EOF
