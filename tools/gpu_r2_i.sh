#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cat > /tmp/dbg.py <<'PY'
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
pb = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
r = optimizer.solve(ctx, pb, optimizer.default_options(ctx.lib, max_iter=1, function_tolerance=1e9))
PY
for st in 0 1 2 3 4; do OV2_BA_BIG=1 OV2_BA_CHOL_STAGES=$st OV2_BA_DUMP=/tmp/big$st.bin python /tmp/dbg.py; done
OV2_BA_BIG=1 OV2_BA_DUMP=/tmp/bigfull.bin python /tmp/dbg.py
python - <<'PY'
import numpy as np
np.set_printoptions(linewidth=220, precision=5)
def load(p):
    a=np.fromfile(p,dtype=np.float64); nfp,nf=int(a[0]),int(a[1]); nn=nfp*nfp
    return a[4+2*nn:4+3*nn].reshape(nfp,nfp)[:nf,:nf], nf
S0,nf=load("/tmp/big0.bin")
A=np.tril(S0)+np.tril(S0,-1).T
print("assembled S symmetric pos def? min eig", np.linalg.eigvalsh(A).min())
L=np.linalg.cholesky(A)
S1,_=load("/tmp/big1.bin"); print("after diag(0): block err", np.nanmax(np.abs(np.tril(S1[:32,:32])-L[:32,:32])), "rest unchanged", np.nanmax(np.abs(np.tril(S1)[32:]-np.tril(S0)[32:])))
S2,_=load("/tmp/big2.bin"); P=S2[32:,:32]; print("after panel(0): panel err", np.nanmax(np.abs(P-L[32:,:32])), "nan", np.isnan(P).sum()); 
if np.isnan(P).any(): print(np.argwhere(np.isnan(P))[:8].tolist()); print("row0 got", P[0,:20], "\nexp", L[32,:20])
S3,_=load("/tmp/big3.bin"); T=np.tril(S3[32:,32:]); Te=np.tril(A[32:,32:]-L[32:,:32]@L[32:,:32].T); print("after trail(0): err", np.nanmax(np.abs(T-Te)))
Sf,_=load("/tmp/bigfull.bin"); print("full: L err", np.nanmax(np.abs(np.tril(Sf)-L)), "nan", np.isnan(np.tril(Sf)).sum())
PY
