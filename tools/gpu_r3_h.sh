python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from exonerate_amd import workloads
n=64
proteins, contig, places = workloads.protein_vs_contig(n, 300, 10000000, seed=20260935, introns=True)
os.makedirs("/tmp/c5p", exist_ok=True)
open("/tmp/c5p/q.fa","w").write("".join(">p%d\n%s\n"%(i,p.decode()) for i,p in enumerate(proteins)))
open("/tmp/c5p/t.fa","w").write(">chr\n%s\n"%contig.decode())
exe="integration/_build/exonerate-gpu"
args=["-m","protein2genome","--showalignment","no","--showvulgar","yes","-V","0","/tmp/c5p/q.fa","/tmp/c5p/t.fa"]
for env in ({}, {"C4GPU_SEED_OFF":"1"}):
    t0=time.perf_counter()
    r=subprocess.run([exe]+args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, C4GPU_TRACE="1", C4GPU_VERBOSE="1", **env))
    print(env, "wall %.2f s"%(time.perf_counter()-t0))
    print("\n".join(l for l in r.stderr.decode().splitlines() if ("sdp" in l or "c4gpu seed" in l or "staging: coded" in l))[:2500])
PY
