"""bench.py's one-GPU pieces that are not the timed region itself."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_scaling_emulated_block_has_every_rank_and_the_exchange_model(gpu):
    """`extra[*].scaling_emulated`: every rank's shard of the 1-D row partition timed on this GPU with
    the schedule spmm_auto would run on it, nnz balance, all-gather bytes, exchange modelled over xGMI -- labelled as an
    emulation.  Small graph: the point is the bookkeeping (ranks cover every edge once, bytes follow the shard pitch)."""
    from isplib_amd import synth
    bench = _bench()
    rowptr, col, n = synth.dataset_like("reddit", device=gpu, scale=0.02)
    rec = bench.scaling_emulated(gpu, rowptr, col, n, 32, "scaled reddit-like", one_gpu_ms=1.0, ranks=(2, 3))
    assert "EMULATED" in rec["what"] and "NO RCCL" in rec["what"]
    assert [p["ranks"] for p in rec["points"]] == [2, 3]
    for p in rec["points"]:
        assert len(p["per_rank"]) == p["ranks"]
        assert sum(r["nnz"] for r in p["per_rank"]) == col.numel()
        assert sum(r["rows"] for r in p["per_rank"]) == n
        assert p["local_spmm_ms_max"] >= p["local_spmm_ms_mean"] > 0
        assert 1.0 <= p["nnz_balance_max_over_mean"] < 1.5
        shard = p["all_gather_bytes_sent_per_rank"]
        assert shard % (192 * 32 * 4) == 0 and shard * p["ranks"] >= n * 32 * 4
        assert p["all_gather_bytes_received_per_rank"] == (p["ranks"] - 1) * shard
        direct = p["exchange_model_ms"]["direct_one_link_per_peer"]
        assert abs(direct - shard / 153e9 * 1e3) < 1e-9 and abs(p["exchange_model_ms"]["ring"] - (p["ranks"] - 1) * direct) < 1e-9
        assert abs(p["step_model_ms"]["direct, no overlap"] - (p["local_spmm_ms_max"] + direct)) < 1e-9
        assert abs(p["compute_speedup_over_one_gpu"] - 1.0 / p["local_spmm_ms_max"]) < 1e-9


def test_one_rank_rccl_rehearsal_prints_exactly_one_line_on_stdout(gpu):
    """The N > 1 code walked by ONE rank over real RCCL (ISPLIB_BENCH_FORCE_DIST=1 under torch.distributed.run).  RCCL writes
    a version banner to file descriptor 1 when its first communicator comes up; rank 0's stdout must still be the one JSON
    line and nothing else (bench.py hands descriptor 1 to stderr and keeps a private copy for the line)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, ISPLIB_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(name, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scale", "0.05"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout[:1000]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and "REHEARSAL" in rec["data"] and rec["value"] > 0
    assert "process group up: backend=nccl" in proc.stderr


def test_default_line_carries_the_contract_fields(gpu):
    """The JSON line of `python bench.py` (one GPU): every field the driver's contract names, the `roofline` and
    `cpu_baseline` objects, and the secondary gather ceiling -- on a scaled graph so that the oracle's leg takes seconds."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ISPLIB_BENCH_FORCE_DIST"):
        env.pop(name, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--scale", "0.05", "--no-extra"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout[:1000]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["metric"] == "edges_aggregated_per_sec" and rec["unit"] == "edges/s" and rec["higher_is_better"] is True
    assert (rec["n_gpus"], rec["steps"], rec["warmup"]) == (1, 4, 2) and rec["vs_baseline"] is None and rec["dtype"] == "f32"
    assert "workload" in rec["config"] and "SCALED" in rec["config"]["workload"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac"] < 1
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    ceil = roof["gather_ceiling"]
    assert ceil["cus"] > 0 and ceil["ms"] > 0 and abs(ceil["frac"] - ceil["ms"] / roof["kernel_avg_ms"]) < 1e-12
    cpu = rec["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["unit"] == "edges/s" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert abs(rec["value"] - int(rec["config"]["workload"].split("nnz=")[1].split(")")[0]) / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]


def test_epoch_emulated_block_times_sampled_ranks_and_models_the_exchange(gpu):
    """config 5's `scaling_emulated` (the partitioned GCN epoch rank by rank on one GPU, all-gathers replaced by local fills):
    bookkeeping on a small graph -- the sampled ranks are the first and the last of every P, their rows / nnz are the
    partition's, the exchange model is six shard transfers on one xGMI link each, and the emulation is labelled as such."""
    from isplib_amd import synth
    from isplib_amd.dist import RowPartition
    bench = _bench()
    rowptr, col, n = synth.dataset_like("reddit", device=gpu, scale=0.02)
    rec = bench.gcn_epoch_config(gpu, rowptr, col, n, with_cpu_epoch=False)
    assert rec["ms"] > 0 and rec["normalize_true_fused"]["ms"] > 0
    emu = rec["scaling_emulated"]
    assert "error" not in emu and "EMULATED" in emu["what"] and "NO RCCL" in emu["what"]
    assert [p["ranks"] for p in emu["points"]] == [2, 4, 8]
    for p in emu["points"]:
        world = p["ranks"]
        assert p["ranks_sampled"] == [0, world - 1] and len(p["per_rank"]) == 2
        for r in p["per_rank"]:
            part = RowPartition(rowptr, col, None, n, r["rank"], world)
            assert (r["rows"], r["nnz"]) == (part.rows, part.nnz) and r["epoch_ms"] > 0
        shard_rows = RowPartition(rowptr, col, None, n, 0, world).max_rows
        model = 3 * (32 + 41) * shard_rows * 4 / 153e9 * 1e3
        assert abs(p["six_all_gathers_model_ms_direct"] - model) < 1e-9
        assert abs(p["epoch_model_ms_no_overlap"] - (p["epoch_compute_ms_max_of_sampled_ranks"] + model)) < 1e-9
