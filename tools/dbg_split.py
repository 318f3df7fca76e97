import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudacam_amd import api, synth
from oracle import oracle as O
O.build()
for (w, h, kind) in [(640, 480, "natural"), (640, 480, "noise"), (248, 100, "natural"), (200, 60, "natural")]:
    img = synth.natural(w, h, 1) if kind == "natural" else synth.noise(w, h, 1)
    want = O.canny_r(img, 10, 40)
    for chunk in (0, 1080, 24):
        with api.Context(w, h, 1, 1) as ctx:
            ctx.set_tuning(chunk, 4)
            got = ctx.process(img)[0]
        bad = np.argwhere(got != want)
        print(w, h, kind, "chunk", chunk, "mismatches", len(bad), "rows", sorted(set(bad[:, 0]))[:12], "cols", sorted(set(bad[:, 1]))[:12])
