"""The XCD-aware work-group map (ov2slam_amd/csrc/xcd_map.hpp, used by the LK, pyramid and CLAHE launches) must be a
bijection ids <-> (item, k): a hole would leave work undone, a collision would do it twice.  The header is plain C++;
it is compiled here with g++ and checked exhaustively for small sizes plus the bench geometries."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <vector>
#include "xcd_map.hpp"
static int check(int per_item, int batch)
{
    const int n = per_item * batch;
    std::vector<int> seen(n, 0);
    for (int id = 0; id < n; id++) {
        int item = -1, k = -1;
        ov2_xcd_map(id, per_item, batch, &item, &k);
        if (item < 0 || item >= batch || k < 0 || k >= per_item) return 1;
        if (seen[item * per_item + k]++) return 2;
        // the point of the map: for the items handled in groups of 8, the id's residue mod 8 is the item's
        if (item < (batch & ~7) && (id & 7) != (item & 7)) return 3;
    }
    return 0;
}
int main()
{
    for (int per_item = 1; per_item <= 40; per_item++)
        for (int batch = 1; batch <= 70; batch++)
            if (int rc = check(per_item, batch)) { printf("FAIL %d %d rc=%d\n", per_item, batch, rc); return 1; }
    const int geo[][2] = {{16, 1024}, {16, 4096}, {45, 4096}, {7, 4096}, {10, 4096}, {20, 1023}, {3, 11}, {135, 3}};
    for (auto &g : geo)
        if (int rc = check(g[0], g[1])) { printf("FAIL %d %d rc=%d\n", g[0], g[1], rc); return 1; }
    printf("OK\n");
    return 0;
}
"""


def test_xcd_map_is_a_bijection(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "ov2slam_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout + out.stderr
