"""csrc/tile_geometry.h — where the SpMM exports cut a column-major result for their tiled downloads — compiled for the
CPU and checked over random geometries (tests/cpp/tile_geometry_check.cpp): page-aligned pieces, every tile inside one
registered piece, the groups' last partial pages in the next piece, exact coverage."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_geometry_properties(tmp_path):
    exe = str(tmp_path / "tile_geometry_check")
    src = os.path.join(ROOT, "tests", "cpp", "tile_geometry_check.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-fsanitize=undefined", "-fno-sanitize-recover", src, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert r.returncode == 0 and "tile geometry ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
