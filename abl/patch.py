import sys
v, path = sys.argv[1], sys.argv[2]
s = open(path).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b)
if v == "nostore":      # transposes kept, interior global stores skipped
    rep("                if (HALF_PIECES % 64 == 0 || lane + 64 * j < HALF_PIECES) *(f32x4 *)(hb + j * 1024) = v[j];",
        "                if ((HALF_PIECES % 64 == 0 || lane + 64 * j < HALF_PIECES) && v[j].x == 12345.678f) *(f32x4 *)(hb + j * 1024) = v[j];")
elif v == "notrans":    # no transposes, no stores
    rep("        if (!bad) store_tile<C, false>(lds, t, (unsigned char *)a.out + (long)s * a.out_stride * OB, tile_e0, lo, 0, pc);",
        "        if (!bad && pc[0].x == 12345.678f && pc[PL - 1].y == 2.5f) store_tile<C, false>(lds, t, (unsigned char *)a.out + (long)s * a.out_stride * OB, tile_e0, lo, 0, pc);")
elif v == "nosecond":   # resampler FMAs skipped
    rep("                for (int i = 0; i < KP; i++) fma2<C::PK>(sacc, YY(b - i), rsv[p + i * L]);",
        "                for (int i = 0; i < 1; i++) fma2<C::PK>(sacc, YY(b - i), rsv[p + i * L]);")
elif v == "fir1":       # FFA middle blocks: 1 of 3
    rep("    for (int b = 1; b < TH / RH; b++) {\n        LOAD_DBLOCK(b)", "    for (int b = 1; b < 2; b++) {\n        LOAD_DBLOCK(b)")
elif v == "noload":     # no HBM reads of the next tile
    rep("            tile_issue_loads<C, KIND>(regs, inn, (long)tn * C::TILE_IN - C::HALO, t);\n        }",
        "            if (a.n_in < 0) tile_issue_loads<C, KIND>(regs, inn, (long)tn * C::TILE_IN - C::HALO, t);\n        }")
elif v == "fakestore":  # same store instructions and addresses, no LDS transposes (data lands in the wrong place)
    a = s.index("#pragma unroll\n    for (int h = 0; h < 2; h++) {\n        if ((lane >> 5) == h) {")
    b = s.index("// wave-uniform: do the chunks this tile reads all have sync offset 0?")
    body = """    if (!CHECKED) {
        unsigned char *hb0 = outb + (tile_e0 + (long)NOUT * (wave * 64)) * OB + lane * 16;
#pragma unroll
        for (int k = 0; k < PL; k++) *(f32x4 *)(hb0 + k * 1024) = pc[k];
        return;
    }
"""
    s = s[:a] + body + s[a:]
elif v == "base":
    pass
open(path, "w").write(s)
