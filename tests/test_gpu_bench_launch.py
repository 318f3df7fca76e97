"""The N > 1 launch path of bench.py on a one-GPU box: the driver starts the multi-GPU
bench as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`, one rank per GPU, RCCL
for the barrier and the MAX-reduce.  No 8-GPU node is available to the builder, so this test runs that exact command
with two ranks as FRESH CHILD PROCESSES (never an exec from a process that touched the GPU), both mapped to device 0 by
HC_BENCH_LOCAL_DEVICE, and checks the one JSON line rank 0 prints: rendezvous on 127.0.0.1, process group of two, frame
blocks per rank, barrier + MAX-reduce, whole-job frames/s.

RCCL may refuse two ranks on one physical device ("Duplicate GPU detected"): a communicator of two ranks needs two
devices.  In that case the same command is run with the gloo backend (HC_BENCH_BACKEND) -- everything but the RCCL
transport is then still the code the 8-GPU run takes -- and the test says which leg it proved.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(backend):
    env = dict(os.environ)
    env["HC_BENCH_LOCAL_DEVICE"] = "0"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("HC_BENCH_BACKEND", None)
    if backend:
        env["HC_BENCH_BACKEND"] = backend
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--rotate", "2", "--unique", "6", "--width", "640", "--height", "480", "--no-cpu-baseline", "--no-host-fed"]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def _json_lines(stdout):
    out = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def test_two_ranks_through_torchrun_on_one_device():
    r = _launch(None)   # the driver's command: backend "nccl" (= RCCL)
    proved = "nccl"
    if r.returncode != 0:
        text = r.stdout + r.stderr
        if "uplicate GPU" not in text and "invalid usage" not in text.lower() and "ncclInvalidUsage" not in text:
            pytest.fail("torchrun --nproc-per-node 2 bench.py failed for a reason other than two ranks sharing a device:\n" + text[-3000:])
        r = _launch("gloo")
        proved = "gloo"
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, f"rank 0 prints ONE JSON line, got {len(lines)}:\n{r.stdout[-2000:]}"
    j = lines[0]
    assert j["n_gpus"] == 2 and j["config"]["world_size"] == 2 and j["config"]["backend"] == proved
    assert j["scaling"] == "weak" and j["config"]["sharding"] == "frames x2"
    assert j["steps"] == 3 and j["value"] > 0
    # whole-job aggregate: both ranks' frames over the MAX-reduced time
    assert abs(j["value"] - 2 * 8 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 0.02
    print(f"N > 1 launch path proved with backend {proved}: {j['value']:.0f} frames/s over 2 ranks on one device")
