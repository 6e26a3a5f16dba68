"""Every MIDAS_* environment switch of the library that chooses between two implementations of the same thing: the alternative
gives the same bits as the default on one fixed scenario through all engine forms (tests/knob_case.py, a fresh process per switch -
the library reads most of them once).  Switches that have tests of their own elsewhere (MIDAS_TAIL_GROUPED, MIDAS_DENSE_ROWS,
MIDAS_DENSE_SCORES, MIDAS_HOST_INDEX, MIDAS_TOPN_STREAM, MIDAS_ANNEAL_SMALL, MIDAS_SCRATCH_LOG) are run here as well where the scenario
reaches them; MIDAS_ABLATE is a profiling switch that changes results by design and is not a product path.  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))

# switch -> value that selects the non-default path (particles.hip, resample.hip, loop.hip, index_build.hip, api.hip)
SWITCHES = [
    ("MIDAS_TAIL_GROUPED", "0"),     # step tail: 25 workgroups instead of 391 waves with hand-over records
    ("MIDAS_TB2_TAB", "1"),          # flush: tables in LDS
    ("MIDAS_WAVE_TABLES", "0"),      # front: workgroup-wide resample tables instead of per-wave ones
    ("MIDAS_SPLIT_FRONT", "0"),      # front: one kernel form for every size
    ("MIDAS_FRONT_WAVES", "2"),      # front: waves per workgroup
    ("MIDAS_FRONT_SMALL", "0"),      # loop: the large-set front for small sets too
    ("MIDAS_LIST_WAVES", "0"),       # sparse scoring without the prediction list's streaming workgroups
    ("MIDAS_PREF_PLAIN", "0"),       # front: prefetch form
    ("MIDAS_PRESORT", "0"),          # batch front: no presort
    ("MIDAS_PRESORT_FUSED", "0"),    # ... presort as two kernels
    ("MIDAS_PRESORT_RUN", "2"),      # ... run length of the deal
    ("MIDAS_PRESORT_CHUNK", "1024"),  # ... slots per presort workgroup
    ("MIDAS_NO_VSCR", "1"),          # prune without the float32 screening records
    ("MIDAS_MESH_FIELD", "0"),       # prune without the distance field
    ("MIDAS_LOOP_MERGE", "0"),       # loop: weights and cluster moments as two launches
    ("MIDAS_MOMENTS_SKIP", "0"),     # cluster moments: every cluster summed by every workgroup, members or not
    ("MIDAS_OVERLAP", "0"),          # no fused front at all: scoring, then the particle update
    ("MIDAS_LAZY_MODULES", "1"),     # kernels loaded on first use
    ("MIDAS_DENSE_SCORES", "1"),     # every codebook row scored by the front's streaming waves
    ("MIDAS_DENSE_ROWS", "1500"),    # ... only in frames whose prediction list is long
    ("MIDAS_HOST_INDEX", "1"),       # neighbour / vertex lists built on the host
]


def _run(env_extra):
    env = dict(os.environ)
    for k, _ in SWITCHES:
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "knob_case.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = {ln.split()[0]: ln for ln in r.stdout.splitlines() if ln.split() and ln.split()[0] in ("pipelined", "eager", "batch", "loop")}
    want = {"pipelined", "eager", "batch", "loop"} - ({"batch"} if "MIDAS_DENSE_SCORES" in env_extra else set())  # (the batch engine is sparse-only)
    assert set(lines) == want, r.stdout[-2000:]
    return lines


@pytest.fixture(scope="module")
def default_lines():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _run({})


def test_scenario_is_reproducible(default_lines):
    assert _run({}) == default_lines


@pytest.mark.parametrize("name,value", SWITCHES, ids=[f"{k}={v}" for k, v in SWITCHES])
def test_switch_changes_no_bit(default_lines, name, value):
    got = _run({name: value})
    assert got == {k: default_lines[k] for k in got}


def test_shard_run_without_the_folded_unpack():
    """MIDAS_SHARD_FOLD=0 (midas_shard_run: every frame unpacks into the particle arrays instead of handing its inbox to the next
    front): the library-driven sharded run's own parity test, in a process that has the switch set."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MIDAS_SHARD_FOLD="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_dist.py"), "-q", "-m", "gpu", "-x",
                        "-k", "single_rank_process_group_nccl"], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
