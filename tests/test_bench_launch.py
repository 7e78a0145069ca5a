"""bench.py's N > 1 entry: `python bench.py --gpus N` starts the ranks itself, a WORLD_SIZE that disagrees with --gpus is
refused, and (GPU box) the two-rank rehearsal on one GPU runs the sharded stream / work-queue legs and gathers the same
results as one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_gpus_flag_without_launcher_starts_ranks_and_propagates_failure():
    """No GPU here: the two child ranks start (torch.distributed.run banner) and fail loudly; bench.py must exit non-zero
    instead of silently measuring one device."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = _run(["--gpus", "2", "--steps", "1"], timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in (r.stderr + r.stdout)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_two_rank_rehearsal_matches_one_rank():
    """MARIE_BENCH_REHEARSE=1 python bench.py --gpus 2 on one GPU (gloo for the start-up broadcast / gathers, both ranks on
    device 0): n_gpus is 2, the weight broadcast checksum passed, and the stream / mixed-DPI results gathered from two ranks
    equal the one-rank results (order-sensitive checksums)."""
    common = ["--steps", "1", "--warmup", "1", "--pages", "4", "--det-batch", "4", "--decode-len", "3", "--stream-pages", "12",
              "--mixed-pages", "24", "--no-cpu-baseline", "--no-kernel-timing", "--no-secondary", "--host-steps", "0"]
    one = _json_line(_run(["--gpus", "1"] + common).stdout)
    r = _run(["--gpus", "2"] + common, {"MARIE_BENCH_REHEARSE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    two = _json_line(r.stdout)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["stream"]["pages"] == one["stream"]["pages"] == 12
    assert two["stream"]["pages_per_rank"] == 6 and two["stream"]["scaling"] == "strong"
    assert two["stream"]["result_checksum"] == one["stream"]["result_checksum"]
    # the mixed-DPI leg processes 24 pages per rank: compare the first 24 pages' structure through the checksum of a 1-rank
    # run over the same 48 pages
    one48 = _json_line(_run(["--gpus", "1"] + [a if a != "24" else "48" for a in common]).stdout)
    assert two["mixed_dpi"]["pages"] == one48["mixed_dpi"]["pages"] == 48
    assert two["mixed_dpi"]["result_checksum"] == one48["mixed_dpi"]["result_checksum"]


@pytest.mark.gpu
def test_rccl_path_with_one_rank_under_the_launcher():
    """The driver's N > 1 command line with N = 1: ``python -m torch.distributed.run --nproc-per-node=1 bench.py --gpus 1``.
    The process group is RCCL (backend nccl, device_id given): the weight arenas go through ``dist.broadcast`` of device
    buffers and the checksum ``all_gather_object``, the stream leg's ``gather_in_order``, the mixed-DPI leg's store-backed work
    queue and the ``all_reduce(MAX)`` of the timings all run on it, and ``destroy_process_group`` ends it cleanly.  The results
    (order-sensitive checksums) equal the launcher-less run's."""
    import socket

    common = ["--gpus", "1", "--steps", "1", "--warmup", "1", "--pages", "4", "--det-batch", "4", "--decode-len", "3", "--stream-pages", "12",
              "--mixed-pages", "24", "--no-cpu-baseline", "--no-kernel-timing", "--no-secondary", "--host-steps", "0"]
    plain = _json_line(_run(common).stdout)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MARIE_BENCH_REHEARSE"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH] + common, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1
    assert line["process_group"] == {"backend": "nccl", "world": 1, "arena_checksums_equal": True}
    assert line["stream"]["pages"] == 12 and line["stream"]["result_checksum"] == plain["stream"]["result_checksum"]
    assert line["mixed_dpi"]["result_checksum"] == plain["mixed_dpi"]["result_checksum"]
    assert "process_group" not in plain


@pytest.mark.gpu
def test_all_legs_run_and_the_process_exits_cleanly():
    """The default command's shape at a reduced size with every secondary leg on (engine_api, overlay, det_passes_3, mixed DPI,
    cpu parity sample off): one JSON line with the contract's fields, exit code 0.  Regression: the engine leg's detector thread
    used to leave its (dead) torch stream in the detector's context, and the teardown's hipStreamSynchronize on it aborted the
    interpreter AFTER the line was printed (exit 134)."""
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--pages", "8", "--det-batch", "4", "--decode-len", "3", "--mixed-pages", "6",
              "--no-cpu-baseline", "--host-steps", "0", "--engine-page-batch", "4"])
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    line = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["value"] > 0
    eng = line["engine_api"]
    assert eng["fixed_lines"]["value"] > 0 and eng["fixed_lines"]["page_batch"] == 4
    assert eng["fixed_lines"]["lines_per_page"] == 40
    assert line["overlay"]["value"] > 0


def test_symbol_error_rate_is_levenshtein_over_reference_length():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._edit_rate([[1, 2, 3, 4]], [[1, 2, 4]]) == 0.25                 # one deletion
    assert bench._edit_rate([[1, 2], [3]], [[1, 2], [3]]) == 0.0
    assert bench._edit_rate([[1, 2, 3]], [[9, 9, 9, 9]]) == pytest.approx(4 / 3)  # three substitutions + one insertion
    assert bench._edit_rate([[5, 6]], [[]]) == 1.0
