cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_chain2; mkdir -p $O
DBAT_HIP_DF_CHAIN=0 DBAT_AMD_LIB=prof DBAT_HIP_DF_TRACE=$O/trace_legacy.csv python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2>$O/err_legacy.txt
python - <<'PY'
import numpy as np
rows=np.loadtxt('gpurun_out/r05_chain2/trace_legacy.csv',delimiter=',',dtype=np.int64)
d=rows[(rows[:,1]==rows[:,2])&(rows[:,1]>=0)]
T=d[:,3:19].astype(float)
print('legacy diagonal tasks: %d' % len(d))
print('potf2 (stamp 2 -> stamp 12, us): median %.2f' % np.median((T[:,12]-T[:,2])*0.01))
print('potf2 cycles (14->15): median %.0f  => %.2f GHz' % (np.median(T[:,15]-T[:,14]), np.median(T[:,15]-T[:,14])/np.median((T[:,12]-T[:,2])*0.01)/1e3))
seq=np.stack([T[:,2],T[:,6],T[:,7],T[:,8],T[:,9],T[:,10],T[:,11],T[:,12]],axis=1)
print('panels/trails', np.median(np.diff(seq,axis=1),axis=0)*0.01)
print('task start->potf2 start %.2f, potf2 end -> task end %.2f' % (np.median((T[:,2]-T[:,0])*0.01), np.median((T[:,4]-T[:,12])*0.01)))
PY
