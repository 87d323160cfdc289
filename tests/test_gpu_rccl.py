"""RCCL on the GPU box: torch.distributed backend "nccl" (= RCCL) with a world of one rank -- all a one-GPU
box allows -- drives the sharded paths with their collectives really issued; child process tests/rccl_world1.py so that
the process group is the first thing that touches the GPU.  And bench.py through the same branch (SP_BENCH_FORCE_DIST=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_world1_sharded_paths():
    d = _run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1.py")])
    assert d["backend"] == "nccl" and d["world"] == 1
    assert d["pipeline_steps"] == 5
    assert d["pipeline_vs_single_call"] <= 1.0          # units of rtol 2e-4 |ref| + 1e-6 max
    assert d["sharded_vs_oracle"] <= 1.0
    assert d["csd_compact_vs_local"] <= 2e-7            # one complex64 rounding of a float32-accurate matrix
    assert d["csd_full_vs_local"] <= 1e-12
    assert d["csd_compact_vs_oracle"] <= 2e-4
    # the streaming engine without a communicator: epilogue on the library's stream beside the next main kernel
    assert d["stream_steps"] == 10 and d["stream_vs_single_call"] <= 1.0 and d["stream_real_and_kaiser"] <= 1.0
    # the collective issued by libspectral on its own stream (sp_comm_init, sp_welch_dist_submit / _flush)
    assert d["native_comm"] == [1, 0] and d["native_comm_info"] == [1, 0]
    assert d["native_pipeline_steps"] == 5
    assert d["native_pipeline_vs_single_call"] <= 1.0
    assert d["native_kaiser_vs_single_call"] <= 1.0
    # main kernel partitioned over ncu - 4 CUs (what a communicator of more than one rank does: room for RCCL's kernel)
    assert d["native_reserved_steps"] == 5 and d["native_reserved_cus_vs_single_call"] <= 1.0


def test_bench_force_dist_world1():
    """bench.py's sharded branch (WelchPipeline over an RCCL group) on one GPU: the JSON line carries both parity gates"""
    d = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--settle-steps", "2",
              "--log2n", "24"], {"SP_BENCH_FORCE_DIST": "1"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and "error" not in d
    assert d["parity"]["prefix_vs_oracle"]["ok"] and d["parity"]["sharded_vs_single_gpu"]["ok"]
    assert "all-reduce" in d["config"]["parallelism"] and "libspectral" in d["config"]["parallelism"]
    # and with torch.distributed carrying the state
    d2 = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--settle-steps", "2",
               "--log2n", "24"], {"SP_BENCH_FORCE_DIST": "1", "SP_BENCH_NATIVE_COMM": "0"})
    assert d2["value"] > 0 and "error" not in d2 and "torch.distributed" in d2["config"]["parallelism"]


def test_bench_two_ranks_one_gpu_strong_and_weak():
    """bench.py --gpus 2 on the one GPU of the box (SP_BENCH_ONE_GPU=1: both ranks on cuda:0, gloo carries the state -- RCCL
    refuses two ranks per device): the strong-scaling split of ONE stream is the headline, the weak figure an extra key, and
    both parity gates (prefix against the CPU oracle, sharded against one GPU's PSD of the whole stream) hold at N = 2"""
    d = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--settle-steps", "1",
              "--log2n", "23", "--gate-log2n", "20"], {"SP_BENCH_ONE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and "error" not in d
    assert d["config"]["total_samples"] == 1 << 23 and d["config"]["samples_per_gpu"] <= (1 << 22) + 4096
    assert d["parity"]["prefix_vs_oracle"]["ok"] and d["parity"]["sharded_vs_single_gpu"]["ok"]
    assert d["weak_scaling"]["samples_per_gpu"] >= 1 << 23 and d["weak_scaling"]["value"] > 0
