#!/usr/bin/env python3
"""Diagnostics: sweeps / active tiles per hysteresis launch on the bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudacam_amd import api, synth
api.preload_hip_runtime()
import torch
import numpy as np
W, H, B = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
uniq = synth.frames("natural", W, H, 8)
d_in = torch.from_numpy(uniq).cuda().repeat(B // 8, 1, 1).contiguous()
d_out = torch.empty_like(d_in)
ctx = api.Context(W, H, 1, B)
ctx.set_option(api.OPT_TEST_HYST_DIAG, 1)
ctx.set_tuning(0, K)
ctx.run_device(d_in.data_ptr(), W, W * H, d_out.data_ptr(), W, W * H, B)
ctx.sync()
print("stats (sweeps_sum, sweeps_max, active_tiles) per launch:", ctx.hysteresis_stats(K))
print("info", ctx.hysteresis_info())
