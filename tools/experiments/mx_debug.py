"""Where does k_front_mx's blur / plane tap differ from the oracle?  Prints mismatch histograms by row and by strip column."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cudacam_amd import api, synth
api.preload_hip_runtime()
from oracle import oracle as O
O.build()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
kind = sys.argv[3] if len(sys.argv) > 3 else "natural"
img = synth.natural(w, h, 1) if kind == "natural" else synth.noise(w, h, 2)
st = O.canny_r(img, 10, 40, stages=True)
with api.Context(w, h, 1, 1) as ctx:
    ctx.set_option(api.OPT_FRONT_MX, 1)
    ctx.set_option(api.OPT_DEBUG_TAPS, 1)
    got = ctx.process(img)[0]
    print("form", ctx.last_run_info())
    blur, thr = ctx.debug_tap(api.TAP_BLUR)[0], ctx.debug_tap(api.TAP_THRESH)[0]
for name, a, b in (("blur", blur, st["blur"]), ("thresh", thr, st["thresh"]), ("edges", got, st["edges"])):
    bad = a != b
    print(f"{name}: {bad.sum()} of {bad.size} differ")
    if not bad.any():
        continue
    rows = bad.sum(axis=1)
    print("  rows with mismatches:", np.flatnonzero(rows)[:40].tolist(), "... counts", rows[np.flatnonzero(rows)[:20]].tolist())
    cols = bad.sum(axis=0)
    cs = np.flatnonzero(cols)
    print("  columns with mismatches (col, col%216, (col%216+2)%28, count):", [(int(c), int(c % 216), int((c % 216 + 2) % 28), int(cols[c])) for c in cs[:60]])
    p = np.argwhere(bad)[:12]
    print("  first:", [(tuple(int(v) for v in q), int(a[tuple(q)]), int(b[tuple(q)])) for q in p])
    if name == "blur":
        d = (a.astype(int) - b.astype(int))[bad]
        vals, cnt = np.unique(d, return_counts=True)
        print("  differences (hip - oracle):", dict(zip(vals.tolist()[:30], cnt.tolist()[:30])))
bad = thr != st["thresh"]
print("thresh mismatches by row % 16:", np.bincount(np.argwhere(bad)[:, 0] % 16, minlength=16).tolist())
print("thresh mismatches by row // 16 (first 12):", np.bincount(np.argwhere(bad)[:, 0] // 16)[:12].tolist())
print("thresh mismatches by (col % 216) % 28:", np.bincount((np.argwhere(bad)[:, 1] % 216) % 28, minlength=28).tolist())
print("thresh mismatches by col // 216:", np.bincount(np.argwhere(bad)[:, 1] // 216).tolist())
kinds, cnt = np.unique(np.stack([thr[bad], st["thresh"][bad]], 1), axis=0, return_counts=True)
print("(hip, oracle) pairs:", [(tuple(int(v) for v in k), int(c)) for k, c in zip(kinds, cnt)])
