import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from ser_amd.engine import get_engine
eng = get_engine()
rng = np.random.default_rng(3)
emb = (rng.standard_normal((57, 187)) * 30).astype(np.float32)
lo = np.array([0, 5, 10, 20], np.int32); hi = np.array([12, 17, 22, 57], np.int32)
out = eng.pool_windows(emb, lo, hi, mode=0).cpu().numpy()
for w in range(4):
    sel = emb[lo[w]:hi[w]].astype(np.float64)
    m = sel.mean(axis=0); s = sel.std(axis=0)
    var = ((sel - m) ** 2).sum(axis=0) / sel.shape[0]
    print(w, "mean mism", (out[w, :187] != m).sum(), "std mism", (out[w, 187:] != s).sum(),
          "std^2 vs var max rel", np.max(np.abs(out[w,187:]**2 - var)/var))
out1 = eng.pool_windows(emb, lo, hi, mode=1).cpu().numpy()
for w in range(4):
    m = emb[lo[w]:hi[w]].mean(axis=0)
    print(w, "f32 mean mism", (out1[w] != m.astype(np.float64)).sum())
