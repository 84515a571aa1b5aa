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

l = 1
a = ext.level(l, 0, padded=True).astype(int); b = ora.level(l, padded=True).astype(int)
d = np.argwhere(a != b)
print("x mod 4 histogram (padded x):", np.bincount(d[:, 1] % 4, minlength=4))
print("y mod 8 histogram:", np.bincount(d[:, 0] % 8, minlength=8))
for (y, x) in d[:12]:
    print((y, x), "got", a[y, x], "exp", b[y, x], "exp row above/below", b[y-1, x], b[y+1, x], "exp left/right", b[y, x-1], b[y, x+1])
# does got match the oracle at another row?
yy = d[0][0]
for dy in range(-3, 4):
    print("row", yy, "vs oracle row", yy + dy, "equal fraction", (a[yy, 19:-19] == b[yy + dy, 19:-19]).mean())
