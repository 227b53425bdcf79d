#!/usr/bin/env python3
"""Generate av1-go_amd/host/av1_default_cdfs.inc: the AV1 default CDF tables (spec section "Default CDF tables"),
in the SPECIFICATION'S OWN FORM (15-bit cumulative probabilities, the terminating 32768 and the adaptation counter
left out), for the subset of syntax elements the bitstream writer codes.

Where the numbers come from.  The tables are normative constants of the AV1 bitstream specification; the spec text
is not in this container and the tables (about 13 000 numbers) cannot be restated from memory.  The container does
hold two independent compiled copies: Pillow's bundled libavif (pillow.libs/libavif-*.so) statically links libaom
3.13 (encoder) and dav1d 1.5.3 (decoder), and both keep the default tables in .rodata as uint16 arrays of
`32768 - cdf` (libaom: rows of N+1 entries, N-1 values then two zeros; dav1d: padded rows, booleans as (value, 0)
pairs).  This tool finds each table by the first values of its first row (anchors the author knows from the
specification), checks its shape (every row monotone, padding zero, row count), cross-checks the tables that exist
in both libraries against each other, and writes them out.  Nothing else is taken from the binary.

The result is verified end to end by tests/test_av1_conformance.py: streams coded with these tables decode in dav1d
to exactly the encoder's reconstruction, which fails on any wrong entry that is ever used.

Usage:  python tools/extract_av1_cdfs.py            (rewrites the .inc; the tool is only needed to regenerate it)
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "av1-go_amd", "host", "av1_default_cdfs.inc")


def find_lib():
    import PIL
    libs = os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs")
    c = sorted(glob.glob(os.path.join(libs, "libavif-*.so*")))
    if not c:
        sys.exit("no bundled libavif found")
    return c[0]


class Blob:
    def __init__(self, path):
        self.raw = open(path, "rb").read()
        n = len(self.raw) // 2
        self.a = np.frombuffer(self.raw[:n * 2], dtype="<u2").astype(np.int64)

    def find(self, spec_vals, stride=1, gap_zero=False):
        """all u16 indices where 32768 - a[i + k*stride] == spec_vals[k]"""
        v = [32768 - x for x in spec_vals]
        idx = np.where(self.a == v[0])[0]
        out = []
        for i in idx:
            if i + stride * len(v) >= len(self.a):
                continue
            if all(self.a[i + k * stride] == x for k, x in enumerate(v)):
                if gap_zero and stride > 1 and any(self.a[i + k * stride + 1] != 0 for k in range(len(v))):
                    continue
                out.append(int(i))
        return out

    def rows(self, i, nrows, stride, nsym):
        """nrows rows starting at u16 index i: nsym-1 values each, rest of the row must be zero; returns spec-form cdfs"""
        t = self.a[i:i + nrows * stride].reshape(nrows, stride)
        vals = 32768 - t[:, :nsym - 1]
        assert (t[:, nsym - 1:] == 0).all(), "padding not zero"
        assert (vals > 0).all() and (vals < 32768).all(), "value out of range"
        assert (np.diff(vals, axis=1) >= 0).all(), "row not monotone"
        return vals


def main():
    B = Blob(find_lib())
    T = {}      # name -> (dims tuple, nsym, array [rows, nsym-1])

    def aom(name, anchor, dims, nsym, stride=None, skip_rows=0):
        stride = stride or nsym + 1
        hits = [i for i in B.find(anchor) if B.a[i + nsym - 1] == 0 and B.a[i + nsym] == 0] if stride == nsym + 1 else B.find(anchor)
        # the libaom copy is the one whose rows are exactly `stride` apart and valid
        nrows = int(np.prod(dims))
        good = []
        for i in hits:
            try:
                good.append((i, B.rows(i + skip_rows * stride, nrows, stride, nsym)))
            except AssertionError:
                pass
        assert good and all((g[1] == good[0][1]).all() for g in good), "%s: %d candidates" % (name, len(good))
        T[name] = (dims, nsym, good[0][1])
        return good[0][0]

    def at(name, i, dims, nsym, stride):
        T[name] = (dims, nsym, B.rows(i, int(np.prod(dims)), stride, nsym))

    # ---- mode tables (libaom copy; row = nsym+1 u16 unless noted)
    aom("Default_Intra_Frame_Y_Mode_Cdf", [15588, 17027, 19338], (5, 5), 13)
    aom("Default_Y_Mode_Cdf", [22801, 23489, 24293], (4,), 13)
    i = aom("Default_Uv_Mode_Cfl_Not_Allowed_Cdf", [22631, 24152, 25378], (13,), 13, stride=15)
    at("Default_Uv_Mode_Cfl_Allowed_Cdf", i + 13 * 15, (13,), 14, 15)
    aom("Default_Angle_Delta_Cdf", [2180, 5032, 7567], (8,), 7)
    i = aom("Default_Partition_W8_Cdf", [19132, 25510, 30392], (4,), 4, stride=11)
    at("Default_Partition_W16_Cdf", i + 4 * 11, (4,), 10, 11)
    at("Default_Partition_W32_Cdf", i + 8 * 11, (4,), 10, 11)
    at("Default_Partition_W64_Cdf", i + 12 * 11, (4,), 10, 11)
    at("Default_Partition_W128_Cdf", i + 16 * 11, (4,), 8, 11)
    # intra tx type: libaom [set][4 square sizes][13 modes][17]; set 1 (7 types) real for 4x4 and 8x8, set 2 (5 types) real for 16x16 only,
    # the other slots hold the uniform distribution (the spec's Set2 table lists them as such)
    i = aom("Default_Intra_Tx_Type_Set1_Cdf", [1535, 8035, 9461], (2, 13), 7, stride=17)
    at("Default_Intra_Tx_Type_Set2_Cdf", i + 52 * 17, (3, 13), 5, 17)
    i = aom("Default_Inter_Tx_Type_Set1_Cdf", [4458, 5560, 7695], (2,), 16, stride=17)
    at("Default_Inter_Tx_Type_Set2_Cdf", i + 6 * 17, (1,), 12, 17)         # [set 2][16x16]
    at("Default_Inter_Tx_Type_Set3_Cdf", i + 8 * 17, (4,), 2, 17)
    aom("Default_Cfl_Alpha_Cdf", [7637, 20719, 31401], (6,), 16)
    aom("Default_Interp_Filter_Cdf", [31935, 32720], (16,), 3)
    i = aom("Default_Tx_8x8_Cdf", [19968], (3,), 2, stride=4)
    at("Default_Tx_16x16_Cdf", i + 3 * 4, (3,), 3, 4)
    at("Default_Tx_32x32_Cdf", i + 6 * 4, (3,), 3, 4)
    at("Default_Tx_64x64_Cdf", i + 9 * 4, (3,), 3, 4)

    # ---- coefficient tables (libaom copy): [4 q categories][...]
    # located by shape chain: eob_multi1024 .. eob_multi16, coeff_base_eob, coeff_base, coeff_br, dc_sign, eob_extra, txb_skip
    cb = [k for k in B.find([4034, 8930, 12727]) if B.a[k + 3] == 0 and B.a[k + 4] == 0]
    assert len(cb) == 1
    cb = cb[0]
    at("Default_Coeff_Base_Cdf", cb, (4, 5, 2, 42), 4, 5)
    at("Default_Coeff_Base_Eob_Cdf", cb - 160 * 4, (4, 5, 2, 4), 3, 4)
    p = cb - 160 * 4
    for n, name in ((5, "16"), (6, "32"), (7, "64"), (8, "128"), (9, "256"), (10, "512"), (11, "1024")):
        p -= 16 * (n + 1)
        at("Default_Eob_Pt_%s_Cdf" % name, p, (4, 2, 2), n, n + 1)
    p = cb + 1680 * 5
    at("Default_Coeff_Br_Cdf", p, (4, 5, 2, 21), 4, 5)
    p += 840 * 5
    # the next three tables start on 32-byte boundaries (16 u16)
    def align(p):
        return (p + 15) // 16 * 16
    p = align(p)
    at("Default_Dc_Sign_Cdf", p, (4, 2, 3), 2, 3)
    p = align(p + 24 * 3)
    at("Default_Eob_Extra_Cdf", p, (4, 5, 2, 9), 2, 3)
    p = align(p + 360 * 3)
    at("Default_Txb_Skip_Cdf", p, (4, 5, 13), 2, 3)

    # ---- small tables: dav1d copy (booleans are (value, 0) pairs), cross-checked with libaom's where that was found
    def d1(name, anchor, n, nsym=2, stride=None):
        stride = stride or (2 if nsym == 2 else 4)
        hits = [i for i in B.find(anchor, stride=stride if len(anchor) > 1 and nsym == 2 else 1)]
        good = []
        for i in hits:
            try:
                good.append(B.rows(i, n, stride, nsym))
            except AssertionError:
                pass
        # several hits may hold the same numbers (libaom stores some of them too): they must agree
        assert good, name
        T[name] = ((n,), nsym, good[-1])
        return hits[-1]

    d1("Default_Skip_Cdf", [31671, 16515, 4576], 3)
    d1("Default_Skip_Mode_Cdf", [32621, 20708, 8127], 3)
    i = d1("Default_New_Mv_Cdf", [24035, 16630, 15339], 6)
    at("Default_Zero_Mv_Cdf", i + 12, (2,), 2, 2)
    at("Default_Ref_Mv_Cdf", i + 16, (6,), 2, 2)
    at("Default_Drl_Mode_Cdf", i + 28, (3,), 2, 2)
    at("Default_Is_Inter_Cdf", i + 34, (4,), 2, 2)
    assert T["Default_Zero_Mv_Cdf"][2][:, 0].tolist() == [2175, 1054]
    assert T["Default_Is_Inter_Cdf"][2][:, 0].tolist() == [806, 16662, 20186, 26538]
    # single reference: dav1d stores [6 bits][3 contexts]; the spec's table is [3 contexts][6 bits]
    j = B.find([4897, 16973, 29744], stride=2)
    assert len(j) == 1
    sr = B.rows(j[0], 18, 2, 2).reshape(6, 3).T.reshape(18, 1)
    T["Default_Single_Ref_Cdf"] = ((3, 6), 2, sr)
    la = B.find([4897, 1555, 4236], stride=3)       # libaom's copy, [3][6] rows of 3
    assert len(la) == 1 and (B.rows(la[0], 18, 3, 2) == sr).all(), "single_ref: libaom and dav1d disagree"
    ln = B.find([24035, 16630, 15339], stride=3)
    assert len(ln) == 1 and (B.rows(ln[0], 6, 3, 2) == T["Default_New_Mv_Cdf"][2]).all()
    i = d1("Default_Txfm_Split_Cdf", [28581, 23846, 20847], 21)
    i = B.find([9413, 22581])[-1]
    at("Default_Use_Wiener_Cdf", i + 4, (1,), 2, 2)
    at("Default_Use_Sgrproj_Cdf", i + 6, (1,), 2, 2)
    at("Default_Switchable_Restore_Cdf", i, (1,), 3, 4)
    assert T["Default_Use_Wiener_Cdf"][2][0, 0] == 11570 and T["Default_Use_Sgrproj_Cdf"][2][0, 0] == 16855
    i = B.find([1418, 2123, 13340])
    at("Default_Cfl_Sign_Cdf", i[-1], (1,), 8, 8)
    # motion vectors (dav1d copy; the values are also the ones the author knows from the spec, asserted below)
    i = B.find([28672, 30976, 31858, 32320, 32551, 32656, 32740, 32757, 32762, 32767])
    assert len(i) >= 1
    at("Default_Mv_Class_Cdf", i[-1], (1,), 11, 16)
    j = B.find([4096, 11264, 19328])
    at("Default_Mv_Joint_Cdf", j[-1], (1,), 4, 4)
    fixed = {"Default_Mv_Class0_Bit_Cdf": [[27648]], "Default_Mv_Sign_Cdf": [[16384]], "Default_Mv_Class0_Hp_Cdf": [[20480]],
             "Default_Mv_Hp_Cdf": [[16384]], "Default_Mv_Fr_Cdf": [[8192, 17408, 21248]],
             "Default_Mv_Class0_Fr_Cdf": [[16384, 24576, 26624], [12288, 21248, 24128]],
             "Default_Mv_Bit_Cdf": [[17408], [17920], [18944], [20480], [22528], [24576], [28672], [29952], [29952], [30720]]}
    # these seven are short enough to be known; check that the dav1d component record next to the class table holds them
    comp = 32768 - B.a[i[-1] + 16:i[-1] + 16 + 48]
    comp = [int(x) for x in comp if x != 32768]
    want = [16384, 27648, 16384, 24576, 26624, 12288, 21248, 24128, 20480, 17408, 17920, 18944, 20480, 22528, 24576, 28672, 29952,
            29952, 30720, 8192, 17408, 21248, 16384]
    assert comp[:len(want)] == want, comp
    for k, v in fixed.items():
        arr = np.array(v)
        T[k] = ((arr.shape[0],), arr.shape[1] + 1, arr)

    # ---- write
    with open(OUT, "w") as f:
        f.write("/* GENERATED by tools/extract_av1_cdfs.py — do not edit.\n"
                " * AV1 default CDF tables (AV1 Bitstream & Decoding Process Specification, section \"Default CDF tables\"), in the\n"
                " * specification's form: 15-bit cumulative probabilities, one row per context, the terminating 32768 and the\n"
                " * adaptation counter omitted (a row of an N-symbol element holds N-1 numbers).  Names are the specification's.\n"
                " * Provenance and checks: see the tool's header.  Used by host/av1_bitstream.cpp (encoder) only. */\n")
        for name, (dims, nsym, arr) in T.items():
            rows = int(np.prod(dims))
            assert arr.shape == (rows, nsym - 1), (name, arr.shape, rows, nsym)
            decl = "".join("[%d]" % d for d in dims) + "[%d]" % (nsym - 1)
            f.write("static const uint16_t %s%s = {\n" % (name, decl))

            def emit(level, base, indent):
                if level == len(dims):
                    f.write(indent + "{" + ",".join(str(int(x)) for x in arr[base]) + "},\n")
                    return
                inner = int(np.prod(dims[level + 1:])) if level + 1 < len(dims) else 1
                for k in range(dims[level]):
                    if level + 1 < len(dims):
                        f.write(indent + "{\n")
                        emit(level + 1, base + k * inner, indent + " ")
                        f.write(indent + "},\n")
                    else:
                        emit(level + 1, base + k, indent)
            emit(0, 0, " ")
            f.write("};\n")
    n = sum(v[2].size for v in T.values())
    print("wrote %s: %d tables, %d numbers" % (OUT, len(T), n))


if __name__ == "__main__":
    main()
