import sys, numpy as np
sys.path.insert(0, '.')
import sdslam_amd
from sdslam_amd.synth import make_image
from oracle import oracle as O
cfg = (1000, 1.2, 8, 20)
img = make_image(0)
ext = sdslam_amd.ORBextractor(*cfg, 640, 480, 1)
ora = O.OrbOracle(*cfg)
ext(img); ora.extract(img)
for l in range(8):
    a = ext.level(l, 0, padded=True).astype(int); b = ora.level(l, padded=True).astype(int)
    d = np.argwhere(a != b)
    print(l, a.shape, "mismatch", len(d), (d[:6].tolist(), d[-3:].tolist()) if len(d) else "")
    if len(d):
        ys = np.unique(d[:, 0]); xs = np.unique(d[:, 1])
        print("   rows", ys[:20], "cols range", xs.min(), xs.max(), "maxdiff", np.abs(a - b).max())
