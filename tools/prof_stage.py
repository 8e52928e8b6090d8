import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import exonerate_amd as ex
from exonerate_amd import workloads
pairs = workloads.est2genome_pairs(1024, 1000, 100000)
eng = ex.Engine(0); model = ex.Model("est2genome")
st = ex.Stage(eng, model)
for _ in range(3):
    print(st.load(pairs))
st.close(); eng.close()
