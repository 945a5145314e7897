"""The flatten step's piece rule (spx_layout.h: for_each_piece, the same code the device kernels run) against its
specification on the CPU: tests/piece_cuts_check.cpp, compiled with g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_piece_cuts_tile_the_run_and_bound_the_image(tmp_path):
    exe = str(tmp_path / "piece_cuts_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "piece_cuts_check.cpp")], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "piece cuts ok" in p.stdout
