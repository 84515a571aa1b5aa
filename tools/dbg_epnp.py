import numpy as np, sys
sys.path.insert(0,'.')
import sdslam_amd
from sdslam_amd import synth, capi
from oracle import oracle as O
K=(500.,500.,320.,240.)
sys.path.insert(0,'tests')
from test_oracle_track import _pnp_problem
for seed in range(6):
    T,Xw,uv,_=_pnp_problem(seed, n=12+seed*10, noise=0.5 if seed%2 else 0.0)
    R,t,e=O.epnp(Xw,uv,K)
    Rg,tg,eg=capi.debug_epnp(Xw,uv,K)
    print(seed, len(Xw), 'err', e, eg, 'dR', np.abs(R-Rg).max(), 'dt', np.abs(t-tg).max())
# near-planar surface case
rng=np.random.default_rng(0)
xy=np.stack([rng.uniform(30,610,200), rng.uniform(30,450,200)],1)
Xw=synth.backproject_on_surface(xy)
T=synth.se3_exp((0.02,-0.01,0.015),(0.4,-0.3,0.5))
Xc=Xw@T[:3,:3].T+T[:3,3]
uv=np.stack([500*Xc[:,0]/Xc[:,2]+320, 500*Xc[:,1]/Xc[:,2]+240],1)
for noise in (0.0, 0.3, 1.0):
    uvn = uv + rng.normal(size=uv.shape)*noise
    uvn = uvn.astype(np.float32).astype(np.float64); Xf = Xw.astype(np.float32).astype(np.float64)
    R,t,e=O.epnp(Xf,uvn,K); Rg,tg,eg=capi.debug_epnp(Xf,uvn,K)
    print('surface noise',noise,'err',e,eg,'dR',np.abs(R-Rg).max(),'dt',np.abs(t-tg).max(), 'vs truth', np.abs(t-T[:3,3]).max(), np.abs(tg-T[:3,3]).max())
