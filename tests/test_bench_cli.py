"""bench.py's control flow on the CPU (--emulate: host-emulation engine, gloo): single process and the
2-rank launch the driver uses for N>1 (torch.distributed.run on 127.0.0.1).  The numbers are meaningless;
the contract of the JSON line and the N>1 path (rendezvous, chain exchange, max-over-ranks timing) are not."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_process():
    res = subprocess.run([sys.executable, "bench.py", "--emulate", "--steps", "3", "--warmup", "1"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = _json_line(res.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])


def test_two_ranks():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29671", "bench.py", "--gpus", "2",
                          "--steps", "3", "--warmup", "1", "--emulate"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["chains"] == 2
    # the line says which device every rank computed on (PCI bus id on a GPU box; bench.py refuses to report when two ranks share one)
    assert d["ranks_devices"] == ["emu:rank0", "emu:rank1"] and len(d["instance_devices"]) == 1
    # one chain with its site patterns split over the two ranks, the block sums all-reduced (gloo here, RCCL on GPUs): the total
    # is the unsharded log-likelihood (asserted inside bench.py against an unsharded evaluation / the reference's value)
    ps = d["pattern_sharded"]
    assert "error" not in ps, ps
    assert ps["scaling"] == "strong" and ps["n_gpus"] == 2 and ps["value"] > 0 and ps["lnL_pinned"]
    assert ps["host_synchronisations_per_step"] == 1        # the block sums meet on the device (mbamdReduceLogLikelihood + all-reduce)
    assert abs(ps["lnL"] - ps["lnL_reference_fp64"]) <= 2e-6 * abs(ps["lnL"])
    # the reference's own MPI build on the shim, two ranks (only where the reference binaries were built)
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mb_emu_mpi")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mbamd_mpirun")):
        m = d["mpi_mcmc"]
        assert m["ranks"] == 2 and m["cases"][0]["generations_per_s"] > 0, m
        assert set(m["devices_of_rank"]) == {"0", "1"} and m["distinct_devices"] >= 1, m      # every rank's engine reported its device


def test_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and a clean environment (what a driver that forgets torch.distributed.run would
    type): the script starts its own ranks -- the line says n_gpus == 2 -- instead of measuring one device and reporting success."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--emulate", "--no-mpi"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["config"]["chains"] == 2 and d["ranks_devices"] == ["emu:rank0", "emu:rank1"]


def test_world_size_must_match_gpus():
    """A launcher whose world differs from --gpus (here: one rank for --gpus 2) is refused with a non-zero exit, not measured."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29683")
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--emulate"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in res.stderr and not [l for l in res.stdout.splitlines() if l.startswith("{")]


def test_two_ranks_sharded():
    """--shard: one chain, site patterns over the N devices inside one instance (all children on the emulated device here)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29673", "bench.py", "--gpus", "2", "--shard", "--no-mpi",
                          "--steps", "3", "--warmup", "1", "--emulate"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["chains"] == 1 and "shards" in d["config"]["parallelism"]
    assert len(d["instance_devices"]) == 2 and len(d["ranks_devices"]) == 2       # one child engine per shard, each reporting its device


def test_eight_ranks():
    """The launch the driver uses on an 8-GPU node, emulated (eight processes, gloo, host-emulation engine): chain-parallel replay,
    the pattern-sharded chain and -- where the reference binaries were built -- the MPI-build MCMC on eight ranks; the JSON the
    driver parses, a device per rank, the compact summary at the end of the line."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", "29677", "bench.py", "--gpus", "8",
                          "--steps", "2", "--warmup", "1", "--emulate"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=1500)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _json_line(res.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["chains"] == 8 and d["value"] > 0
    assert d["ranks_devices"] == ["emu:rank%d" % r for r in range(8)]
    ps = d["pattern_sharded"]
    assert "error" not in ps, ps
    assert ps["n_gpus"] == 8 and ps["scaling"] == "strong" and abs(ps["lnL"] - ps["lnL_reference_fp64"]) <= 2e-6 * abs(ps["lnL"])
    assert "timed_out" not in d and list(d)[-1] == "summary" and "columns" in d["summary"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mb_emu_mpi")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mbamd_mpirun")):
        m = d["mpi_mcmc"]
        assert "error" not in m, m
        assert m["ranks"] == 8 and m["cases"][0]["chains"] == 8 and m["cases"][0]["generations_per_s"] > 0, m
        assert set(m["devices_of_rank"]) == {str(r) for r in range(8)}, m


def test_secondary_modes_are_bounded():
    """A secondary mode that does not come back must not cost the scaling line: with a 20 ms budget (the emulated MPI run
    and the pattern-sharded chain taking longer) every rank leaves, rank 0 prints the primary measurement and says what it gave up on."""
    if not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mb_emu_mpi")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mbamd_mpirun"))):
        import pytest
        pytest.skip("needs the reference's MPI build on the shim (oracle/_ref)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29679", "bench.py", "--gpus", "2",
                          "--steps", "2", "--warmup", "1", "--emulate", "--secondary-timeout", "0.02"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d.get("timed_out"), d.get("timed_out")
