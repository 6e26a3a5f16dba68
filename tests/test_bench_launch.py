"""bench.py can be started both ways the driver may start it (contract: `python bench.py --gpus N ...` and
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`): with --gpus 2 and no launcher it launches its
own ranks.  Exercised here through --launch-check (ranks + process group + one collective + ONE JSON line from rank 0; gloo
on this GPU-less machine, RCCL on a GPU box) - the measuring part needs MI355X GPUs and is covered by tests/test_gpu_rccl2.py."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out  # exactly one JSON line on stdout
    return json.loads(lines[0])


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""  # the gloo leg, also on a GPU box
    env["CUDA_VISIBLE_DEVICES"] = ""
    return env


def test_plain_invocation_launches_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["launch_check"] and line["ok"] and line["world"] == 2 and line["n_gpus"] == 2
    assert sorted(rk for rk, _ in line["ranks"]) == [0, 1]


def test_contract_launcher_invocation():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["world"] == 2


def test_too_few_gpus_is_an_error_not_a_line():
    """Without GPUs the measuring invocation exits non-zero and prints no JSON (a wrong world must never yield a line)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True,
                       text=True, timeout=300, env=_env(), cwd=REPO)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
