import sys,time,os; sys.path.insert(0,".")
import exonerate_amd as ex
from exonerate_amd import workloads
QL=int(sys.argv[1]) if len(sys.argv)>1 else 1700
pairs=workloads.est2genome_batches([0],2048,QL,100000)[0]
eng=ex.Engine(0); model=ex.Model("est2genome")
os.environ["C4GPU_TRACE"]="1"
for envv in ("1","0"):
    os.environ["C4GPU_PK16_LONG"]=envv
    b=ex.ResidentBatch(eng,model,pairs); b.run(2)
    for m in range(4): b.kernel_stats(m,reset=True)
    t0=time.perf_counter(); b.run(2); dt=time.perf_counter()-t0
    cells=sum((len(q)+1)*(len(t)+1) for q,t in pairs)
    print("Q %d x 2048 pairs, staged-long %s: %.1f ms  %.3e cells/s  score %.1f region %.1f ckpt %.1f path %.1f" % (QL,envv,dt*1e3,cells/dt,b.kernel_stats(0)["ms"],b.kernel_stats(2)["ms"],b.kernel_stats(3)["ms"],b.kernel_stats(1)["ms"]),flush=True)
    b.close()
