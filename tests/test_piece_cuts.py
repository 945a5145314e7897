"""The flatten step's piece rule (spx_layout.h: for_each_piece, the same code the device kernels run) against its
specification on the CPU: tests/piece_cuts_check.cpp, compiled with g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_piece_cuts_tile_the_run_and_bound_the_image(tmp_path):
    exe = str(tmp_path / "piece_cuts_check")
    # (address + undefined-behaviour sanitizers: a read past the run starts S[0..r] would stop the program)
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(ROOT, "tests", "piece_cuts_check.cpp")], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "piece cuts ok" in p.stdout


def test_balancing_passes_on_the_host_match_what_the_device_printed(tmp_path):
    """The run list of tests/test_gpu_parity.py::test_balanced_pieces_bound_the_walk_past_the_landing_row through a
    host restatement of the passes (tests/balance_passes_host.cpp, the same cut rule): the rows after every pass and
    the longest image before it are the ones the device's passes logged on the MI355X
    (profiles/r03_balanced_pieces_default_build.txt, SPX_TIMING=1)."""
    import numpy as np

    rng = np.random.default_rng(8)
    r = 1 << 16
    idx = np.cumsum(rng.integers(1, 4, size=r)) % 4
    lens = np.minimum((rng.pareto(1.2, size=r) + 1).astype(np.int64), 1 << 18)
    heads = np.frombuffer(b"ACGT", dtype=np.uint8)[idx].copy()
    heads[r // 2], lens[r // 2] = 0, 1
    path = str(tmp_path / "runs.bin")
    with open(path, "wb") as f:
        f.write(np.uint64(r).tobytes())
        f.write(heads.tobytes())
        f.write(lens.astype(np.uint64).tobytes())
    exe = str(tmp_path / "balance_passes_host")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(ROOT, "tests", "balance_passes_host.cpp")], check=True)
    p = subprocess.run([exe, path, "8", "4"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = [tuple(int(x) for x in __import__("re").findall(r"\d+", line)[1:]) for line in p.stdout.splitlines()]
    # (rows before, rows after, longest image before the pass) as logged by the device passes
    assert got == [(65536, 69035, 1393), (69035, 69473, 46), (69473, 69567, 11), (69567, 69585, 9)], p.stdout
    p = subprocess.run([exe, path, "16", "1"], capture_output=True, text=True, timeout=300)
    assert "65536 rows -> 66851; longest image 1393" in p.stdout, p.stdout  # profiles/r03_balanced_pieces_first_run.txt
