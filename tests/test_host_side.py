"""CPU: the host-side C of the product (no GPU needed): table, ratio/configuration scalars, closed forms, sharding,
the polyphase re-indexing, the exported C ABI, and that the resample entry points fail loudly without a device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import _cases
import _checkers as ck
import _product
import clownresampler_amd as cr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    """include/*.h is the contract: every function it declares must be in the .so (all radius instances)."""
    lib = C.CDLL(cr.LIB_PATH)
    names = set()
    for header in ("clownresampler.h", "clownresampler_amd.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(ClownResampler(?:AMD)?_\w+)\s*\(", text))
    names -= {"ClownResamplerAMD_ErrorHandler"}
    assert len(names) >= 30
    per_radius = {n for n in names if n.startswith("ClownResampler_")} | {"ClownResamplerAMD_PlanCreate"}
    for n in sorted(names):
        assert hasattr(lib, n), n
        if n in per_radius:
            assert hasattr(lib, n + "_R8"), n + "_R8"


def test_no_internal_symbols_leak():
    out = subprocess.run(["nm", "-D", "--defined-only", cr.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms and all(s.startswith("ClownResampler") for s in syms), [s for s in syms if not s.startswith("ClownResampler")]


def test_struct_layouts_match_reference_lp64():
    # Configuration 32, LowLevel_State 64, HighLevel_State 8296, Precomputed 49152 (R=3)
    assert C.sizeof(cr.LowestLevel_Configuration) == 32
    assert C.sizeof(cr.LowLevel_State) == 64
    assert C.sizeof(cr.HighLevel_State) == 8296
    assert C.sizeof(cr.load(3).Precomputed) == 49152
    assert C.sizeof(cr.load(8).Precomputed) == 8 * 2 * 1024 * 8
    # field-for-field with the oracle/reference layout used by the checkers
    assert [getattr(cr.LowLevel_State, n).offset for n in ("lowest_level", "channels", "position_integer", "position_fractional", "increment")] == \
           [getattr(ck.LowLevel, n).offset for n in ("cfg", "channels", "pos_int", "pos_frac", "increment")]


def test_header_compiles_as_c89_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#define CLOWNRESAMPLER_IMPLEMENTATION\n#define CLOWNRESAMPLER_STATIC\n#include "clownresampler.h"\n'
                   'int main(void){ClownResampler_Precomputed p; ClownResampler_LowLevel_State s; (void)p; (void)s; return (int)sizeof(ClownResampler_HighLevel_State) != 8296;}\n')
    for cmd in (["gcc", "-std=c89", "-pedantic", "-Wall", "-Werror"], ["g++", "-x", "c++", "-Wall", "-Werror"]):
        exe = tmp_path / "t"
        subprocess.run(cmd + ["-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0
    src8 = tmp_path / "t8.c"
    src8.write_text('#define CLOWNRESAMPLER_KERNEL_RADIUS 8\n#include "clownresampler_amd.h"\n'
                    'int main(void){ClownResampler_Precomputed p; ClownResampler_Precompute(&p); return p.lanczos_kernel_table[8192] != 65536;}\n')
    exe = tmp_path / "t8"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src8), "-o", str(exe), "-L", os.path.dirname(cr.LIB_PATH),
                    "-lclownresampler_amd", "-Wl,-rpath," + os.path.dirname(cr.LIB_PATH)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


# a client that uses the helper macros a reference client may use: the 16.16 conversions (reference clownresampler.h:620-625)
# and the printf conversions of the cc_* types (:503-543 / :562-602), with both integer-type choices
_MACRO_CLIENT = r"""
#include <stdio.h>
#include <string.h>
#include "clownresampler.h"
int main(void)
{
	char text[256];
	const cc_s32f one = CLOWNRESAMPLER_TO_FIXED_POINT_FROM_INTEGER(1);
	const cc_s32f x = CLOWNRESAMPLER_TO_FIXED_POINT_FROM_INTEGER(5) / 2;        /* 2.5 */
	const cc_s32l l32 = -123456789; const cc_u32l ul32 = 4000000000ul; const cc_s32f f32 = -7; const cc_u32f uf32 = 0xABCDEFul;
	const cc_s16l l16 = -1234; const cc_u16f uf16 = 65535u; const cc_s8l l8 = -5; const cc_u8f uf8 = 200u;
	if (one != 65536L || CLOWNRESAMPLER_FIXED_POINT_FRACTIONAL_SIZE != 65536L) return 1;
	if (CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_FLOOR(x) != 2 || CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_ROUND(x) != 3
	 || CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_CEILING(x) != 3 || CLOWNRESAMPLER_TO_INTEGER_FROM_FIXED_POINT_CEILING(one) != 1) return 2;
	if (CLOWNRESAMPLER_FIXED_POINT_MULTIPLY(x, x) != 6 * 65536L + 16384L) return 3;                  /* 6.25 */
	if (CLOWNRESAMPLER_FIXED_POINT_MULTIPLY(-x, one / 65536L * 3) != -7) return 4;                  /* truncation toward zero */
	if (CLOWNRESAMPLER_CLAMP(-3, 3, 7) != 3 || CLOWNRESAMPLER_MIN(2, 1) != 1 || CLOWNRESAMPLER_MAX(2, 1) != 2) return 5;
	sprintf(text, FMT(CC_PRIdLEAST32) " " FMT(CC_PRIuLEAST32) " " FMT(CC_PRIiFAST32) " " FMT(CC_PRIxFAST32) " " FMT(CC_PRIXFAST32) " " FMT(CC_PRIoFAST32),
	        l32, ul32, f32, uf32, uf32, uf32);
	if (strcmp(text, "-123456789 4000000000 -7 abcdef ABCDEF 52746757") != 0) return 6;
	sprintf(text, FMT(CC_PRIdLEAST16) " " FMT(CC_PRIuFAST16) " " FMT(CC_PRIiLEAST8) " " FMT(CC_PRIuFAST8) " " FMT(CC_PRIxFAST8) " " FMT(CC_PRIXLEAST16) " " FMT(CC_PRIoLEAST8),
	        l16, uf16, l8, uf8, uf8, (cc_u16l)0xBEEF, (cc_u8l)8);
	if (strcmp(text, "-1234 65535 -5 200 c8 BEEF 10") != 0) return 7;
	return 0;
}
"""


@pytest.mark.parametrize("c99", [False, True])
def test_header_has_the_references_helper_macros(tmp_path, c99):
    src = tmp_path / "m.c"
    # the C89 names are whole conversion specifications ("%ld"), the C99 ones are <inttypes.h>'s (no "%"): the reference's quirk
    # (and a C99 client of the reference has to bring <inttypes.h> itself: the reference only includes <stdint.h>)
    src.write_text(('#include <inttypes.h>\n#define FMT(x) "%" x\n' if c99 else '#define FMT(x) x\n') + _MACRO_CLIENT)
    flags = ["-std=c99", "-DCC_USE_C99_INTEGERS", "-DCLOWNRESAMPLER_AMD_LINKS_C99_LIBRARY"] if c99 else ["-std=c89", "-Wno-long-long"]
    exe = tmp_path / "m"
    subprocess.run(["gcc"] + flags + ["-pedantic", "-Wall", "-Wno-format", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    if os.path.exists("/root/reference/clownresampler.h"):
        # the same client against the real header, where it exists: same answers
        ref = tmp_path / "r"
        subprocess.run(["gcc"] + flags + ["-Wno-format", "-I", "/root/reference", str(src), "-o", str(ref)], check=True)
        assert subprocess.run([str(ref)]).returncode == 0


def test_c99_integer_abi_is_a_library_of_its_own(tmp_path):
    """CC_USE_C99_INTEGERS (reference clownresampler.h:483-501) changes the widths of the cc_* types: the second build of the
    same sources.  Layouts against the reference compiled the same way (where it exists), and the client-side guard."""
    lib = os.path.join(os.path.dirname(cr.LIB_PATH), "libclownresampler_amd_c99.so")
    assert os.path.exists(lib), "build() makes it (make c99)"
    prog = ('#include <stdio.h>\n#include <stddef.h>\n#include "clownresampler.h"\n'
            'int main(void){printf("%lu %lu %lu %lu %lu %lu %lu\\n", (unsigned long)sizeof(ClownResampler_Precomputed), (unsigned long)sizeof(ClownResampler_LowLevel_State),'
            '(unsigned long)sizeof(ClownResampler_HighLevel_State), (unsigned long)offsetof(ClownResampler_LowLevel_State, position_integer),'
            '(unsigned long)offsetof(ClownResampler_HighLevel_State, input_buffer), (unsigned long)sizeof(cc_s32f), (unsigned long)sizeof(cc_u8f)); return 0;}\n')
    src = tmp_path / "s.c"
    src.write_text(prog)
    exe = tmp_path / "s"
    subprocess.run(["gcc", "-std=c99", "-DCC_USE_C99_INTEGERS", "-DCLOWNRESAMPLER_AMD_LINKS_C99_LIBRARY", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    ours = subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()
    assert ours[0] == str(6144 * 4) and ours[5] == "8" and ours[6] == "1"     # int_least32_t table, long samples, uint_fast8_t channels (glibc x86-64)
    if os.path.exists("/root/reference/clownresampler.h"):
        ref = tmp_path / "sr"
        subprocess.run(["gcc", "-std=c99", "-DCC_USE_C99_INTEGERS", "-I", "/root/reference", str(src), "-o", str(ref)], check=True)
        assert subprocess.run([str(ref)], capture_output=True, text=True).stdout.split() == ours
    # without the acknowledgement the header refuses: the default library has the other layouts
    bad = subprocess.run(["gcc", "-std=c99", "-DCC_USE_C99_INTEGERS", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert bad.returncode != 0 and "libclownresampler_amd_c99" in bad.stderr
    # same public symbols in both builds
    def syms(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    assert syms(lib) == syms(cr.LIB_PATH)


def test_cmake_target_has_the_references_name(tmp_path):
    """`add_subdirectory(<this repo>)` + `target_link_libraries(app clownresampler)`: what a client of the reference's
    CMakeLists.txt (:9, a library target named clownresampler) already has in its build."""
    import shutil
    if shutil.which("cmake") is None:
        pytest.skip("no cmake here")
    app = tmp_path / "app"
    app.mkdir()
    (app / "main.c").write_text('#define CLOWNRESAMPLER_IMPLEMENTATION\n#define CLOWNRESAMPLER_STATIC\n#include "clownresampler.h"\n#include <stdio.h>\n'
                                'static ClownResampler_Precomputed p;\n'
                                'int main(void){ClownResampler_LowLevel_State s; ClownResampler_Precompute(&p);\n'
                                ' if(!ClownResampler_LowLevel_Init(&s, 2, 44100, 48000, 44100)) return 1;\n'
                                ' printf("%lu %ld\\n", (unsigned long)s.increment, (long)p.lanczos_kernel_table[3072]); return 0;}\n')
    (app / "CMakeLists.txt").write_text('cmake_minimum_required(VERSION 3.13)\nproject(app LANGUAGES C)\n'
                                        'add_subdirectory("%s" clownresampler)\nadd_subdirectory("%s" clownresampler_again)\n'
                                        'add_executable(app main.c)\ntarget_link_libraries(app clownresampler)\n' % (ROOT, ROOT))
    build = tmp_path / "b"
    subprocess.run(["cmake", "-S", str(app), "-B", str(build), "-DCMAKE_BUILD_TYPE=Release"], check=True, capture_output=True)
    subprocess.run(["cmake", "--build", str(build)], check=True, capture_output=True)
    out = subprocess.run([str(build / "app")], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.split() == ["60211", "65536"]


@pytest.mark.parametrize("radius", [3, 5, 8])
def test_table_is_the_references(golden, radius):
    # host-side, reference clownresampler.h:955-961
    import hashlib
    t = _product.Product(radius).table()
    assert hashlib.sha256(t.astype("<i4").tobytes()).hexdigest() == golden["table"][str(radius)]["sha256_i32le"]
    assert np.array_equal(t, ck.oracle(radius).table())


@pytest.mark.parametrize("radius", [3, 5, 8])
def test_config_scalars_are_the_references(golden, radius):
    # host-side, reference clownresampler.h:913-984, :1044-1056; failure cases leave exactly what the reference leaves
    p = _product.Product(radius)
    for row in golden["config"][str(radius)]:
        raw = p.api.LowLevel_State()
        raw.position_integer = raw.position_fractional = raw.increment = 0x5A5A5A5A
        ll = raw.lowest_level
        ll.stretched_kernel_radius = ll.integer_stretched_kernel_radius = ll.stretched_kernel_radius_delta = ll.kernel_step_size = 0x5A5A5A5A
        ok = p.api.LowLevel_Init(raw, 2, *row["rates"])
        assert int(ok) == row["ok"], row
        assert [int(v) for v in raw.astuple()] == row["state"], row


def test_highlevel_init_adjust_rules():
    # reference clownresampler.h:1103, :1109-1115, :1188-1206
    p, o = _product.Product(3), ck.oracle(3)
    assert not p.high_init(17, 44100, 48000, 44100)[0]
    for first, then in [((48000, 8000, 8000), (48000, 44100, 44100)), ((44100, 48000, 44100), (48000, 8000, 8000)),
                        ((48000, 24000, 24000), (0, 1, 1)), ((48000, 24000, 24000), (48000, 12000, 12000))]:
        ok, a = p.high_init(2, *first)
        ok2, b = o.high_init(2, *first)
        assert ok == ok2 and (a.lead_needed, a.trail_left, a.max_radius_frames) == (b.lead_needed, b.trail_left, b.max_radius_frames)
        assert p.high_adjust(a, *then) == o.high_adjust(b, *then)
        assert a.low.astuple() == b.low.astuple()


def test_closed_forms_against_oracle_walk():
    import random
    rng = random.Random(7)
    p, o = _product.Product(3), ck.oracle(3)
    for _ in range(300):
        i, out = rng.randrange(1, 200000), rng.randrange(1, 200000)
        ok, a = p.low_init(1, i, out, min(i, out))
        ok2, b = o.low_init(1, i, out, min(i, out))
        if not ok or b.cfg.table_step == 0 or b.cfg.radius_frames > 300:
            continue
        a.raw.position_integer = b.pos_int = rng.randrange(0, 50)
        a.raw.position_fractional = b.pos_frac = rng.randrange(0, 65536)
        frames = rng.randrange(0, 400)
        n = p.api.CountOutputFrames(a.raw, frames)
        assert n == ck.count_output_frames(b, frames)
        padded = ck.pad_frames(ck.noise_pcm(frames), 1, int(b.cfg.radius_frames))
        got, left, ran_out = o.low_resample_i32(b, padded, frames)
        assert len(got) == n
        # state after n frames + exhaustion bookkeeping == AdvanceState then subtract (clownresampler.h:1065-1067)
        p.api.AdvanceState(a.raw, n)
        assert (a.pos_int - frames, a.pos_frac) == (b.pos_int, b.pos_frac)


def test_cfg_counts():
    p = _product.Product(3)
    ok, st = p.low_init(2, 44100, 48000, 44100)
    assert p.api.CountOutputFrames(st.raw, 26460000) == 28800096
    assert p.api.CountOutputFrames(st.raw, 158760000) == 172800574


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_shard_plan_reassembles_the_stream(shards):
    """Sharding is host logic; here each shard is run by the ORACLE (test-only) to prove that the planned
    (pointer offset, frame count, start state) triples concatenate to the one-shot stream bit for bit."""
    o = ck.oracle(3)
    p = _product.Product(3)
    for rates, ch, frames in [((44100, 48000, 44100), 2, 30011), ((48000, 44100, 44100), 8, 9000), ((44100, 8000, 8000), 2, 20000), ((8000, 96000, 8000), 1, 700)]:
        ok, whole = o.low_init(ch, *rates)
        R = int(whole.cfg.radius_frames)
        padded = ck.pad_frames(ck.noise_pcm(frames * ch, 5), ch, R)
        want, _, _ = o.low_resample_i32(whole, padded, frames)
        ok, st = p.low_init(ch, *rates)
        parts, total = [], 0
        for s in range(shards):
            sh = p.api.PlanShard(st.raw, frames, s, shards)
            assert sh.first_output_frame == total and sh.halo_frames == R
            ok, ost = o.low_init(ch, *rates)
            ost.pos_int, ost.pos_frac = sh.state.position_integer, sh.state.position_fractional
            got, left, ran_out = o.low_resample_i32(ost, padded[sh.first_input_frame * ch:], sh.input_frames, capacity=sh.output_frames)
            assert got.size == sh.output_frames * ch, (rates, s)
            # the shard never looks beyond its halo
            assert sh.first_input_frame + sh.input_frames <= frames
            parts.append(got)
            total += sh.output_frames
        assert np.array_equal(np.concatenate(parts), want)


POLY_CONFIGS = [(3, (44100, 48000, 44100)), (3, (48000, 44100, 44100)), (3, (44100, 8000, 8000)), (3, (8000, 44100, 8000)),
                (8, (8000, 96000, 8000)), (8, (48000, 44100, 44100)), (3, (1, 1, 1)), (3, (44100, 48000, 20000)),
                (3, (65537, 65536, 65536)), (3, (1000, 999, 999)), (3, (48000, 44100, 10000)), (3, (3, 7, 2)), (8, (48000, 8000, 8000)),
                (3, (192000, 8000, 8000)), (3, (44100, 1000, 1000))]


@pytest.mark.parametrize("radius,rates", POLY_CONFIGS)
def test_polyphase_rows_equal_reference_taps(radius, rates):
    """For EVERY fractional position: the row the device would pick, laid on the common window, holds exactly the taps
    the reference's definition gives (clownresampler.h:993-1016) and the exact reciprocal (:1025)."""
    p = _product.Product(radius)
    ok, cfg = p.configure(*rates)
    assert ok
    row = np.zeros(65536, dtype=np.uint32)
    info, rows, eligible, why = p.api.BuildRows(cfg, p.pre, row)
    assert rows is not None, why
    table = p.table()
    skr, R, delta, step = cfg.stretched_kernel_radius, cfg.integer_stretched_kernel_radius, cfg.stretched_kernel_radius_delta, cfg.kernel_step_size
    frac = np.arange(65536, dtype=np.int64)
    mr = (frac + delta + 65535) >> 16
    xr = (frac + skr) >> 16
    ks = (step * ((mr << 16) - frac)) >> 16
    taps = R + xr - mr
    if info.row_mode == 1:
        assert np.array_equal(row, (65536 - frac) >> 6)
    assert row.max() == info.rows - 1 and len(np.unique(row)) == info.rows      # every row is used, none beyond
    T, first = info.slots, info.first_slot
    exp = np.zeros((65536, info.row_stride), dtype=np.int64)
    for t in range(int(taps.max())):
        active = t < taps
        w = np.where(active, table[np.minimum(ks + step * t, len(table) - 1)], 0)
        # pure-upsampling rows lie on ONE window common to all phases; affine rows start at their own phase's first tap
        # (cr_plan.c "SHIFTED windows": first_slot is then the window start of the phases with the smallest min_relative)
        slot = mr + t - first if info.row_mode == 1 else t - (first - int(mr.min())) + 0 * mr
        inside = active & (slot >= 0) & (slot < T)
        assert np.all(w[active & ~inside] == 0)      # everything trimmed away is a zero weight
        exp[frac[inside], slot[inside]] = w[inside]
    sums = exp[:, :T].sum(axis=1)
    exp[:, T] = (2 ** 31) // sums                    # 0x80000000 / sum, positive sums (clownresampler.h:1025)
    assert np.all(sums > 0)
    assert np.array_equal(rows[row], exp)
    if eligible:
        assert np.all(np.abs(rows[:, :T]) < 2 ** 23) and np.all(rows[:, T] > 0)


def test_polyphase_shapes_of_baseline_configs():
    p3, p8 = _product.Product(3), _product.Product(8)
    info, rows, el, _ = p3.api.BuildRows(p3.configure(44100, 48000, 44100)[1], p3.pre)
    assert (info.slots, info.first_slot, info.rows, info.row_stride, info.row_mode, el) == (5, 1, 1025, 8, 1, True)   # cfg 2 / 5
    info, rows, el, _ = p8.api.BuildRows(p8.configure(8000, 96000, 8000)[1], p8.pre)
    assert (info.slots, info.first_slot, info.rows, info.row_stride, info.row_mode, el) == (15, 1, 1025, 16, 1, True)  # cfg 3
    info, rows, el, _ = p3.api.BuildRows(p3.configure(48000, 44100, 44100)[1], p3.pre)
    assert (info.slots, info.row_stride, info.row_mode, el) == (6, 8, 0, True)                                       # cfg 4: 5-6 taps, shifted windows


def test_periodic_instance_constants_match_the_rows():
    """k_int's periodic instances (csrc/cr_inst_int_d.hip) are compiled for the window starts, slot signs, 65536-weights and zero
    slots of the reference table's rows at fraction 0; the library checks every launch against them, so a drift would silently
    send those ratios back to the ordinary kernels.  Host-only: what ClownResamplerAMD_PeriodicShape derives from the rows must be
    what stands in the instance table (and what tools/int_shapes.py printed when the instances were written)."""
    src = open(os.path.join(os.path.dirname(cr.__file__), "csrc", "cr_inst_int_d.hip")).read()
    table = set()
    for m in re.finditer(r"make_per<\s*(\d+),\s*(\d+),\s*(\d+),\s*0x([0-9A-Fa-f]+)u,\s*(\d+),\s*(\d+),\s*0x([0-9A-Fa-f]+)ull,\s*0x([0-9A-Fa-f]+)ull(?:,\s*0x([0-9A-Fa-f]+)ull)?>\(\)", src):
        ch, ratio, period, offs, slots, k, neg, safe, zero = m.groups()
        table.add((int(ratio), int(period), int(offs, 16), int(slots), int(neg, 16), int(safe, 16), int(zero or "0", 16)))
    assert len(table) >= 8
    found = set()
    for radius, rates in ((3, (48000, 32000)), (3, (24000, 48000)), (5, (48000, 32000)), (5, (24000, 48000)), (5, (12000, 48000)),
                          (8, (48000, 32000)), (8, (24000, 48000)), (8, (12000, 48000))):
        p = _product.Product(radius)
        ok, st = p.low_init(1, rates[0], rates[1], min(rates))
        sh = p.api.PeriodicShape(st.raw.lowest_level, p.pre, st.raw.increment)
        assert sh is not None and sh["period"] in (2, 4) and sh["starts"][0] == 0
        offs = sum(v << (8 * i) for i, v in enumerate(sh["starts"]))
        key = (sh["ratio"], sh["period"], offs, sh["slots"], sh["negmask"], sh["safemask"], sh["zeromask"])
        assert key in table, (radius, rates, sh)
        found.add(key)
    assert found == table, "an instance in cr_inst_int_d.hip that no tested ratio produces: %s" % (table - found)
    # a ratio whose increment is truncated (1:3: floor(65536 / 3)) never repeats: no period
    p = _product.Product(3)
    ok, st = p.low_init(1, 16000, 48000, 16000)
    assert p.api.PeriodicShape(st.raw.lowest_level, p.pre, st.raw.increment) is None


def test_rejects_what_the_reference_cannot_run():
    p = _product.Product(3)
    ok, cfg = p.configure(2048, 1, 1)      # step underflows to 0: every tap reads table[0] == 0 -> weight sum 0
    assert ok and cfg.kernel_step_size == 0
    info, rows, el, why = p.api.BuildRows(cfg, p.pre)
    assert rows is None and "sum 0" in why


@pytest.mark.skipif(cr.load(3).DeviceCount() > 0, reason="a GPU is present")
def test_resample_fails_loudly_without_a_device():
    """No CPU fallback: with no usable HIP device every resample entry point reports ERROR_NO_DEVICE."""
    p = _product.Product(3)
    ok, st = p.low_init(2, 44100, 48000, 44100)
    padded = ck.pad_frames(ck.noise_pcm(200), 2, 3)
    with pytest.raises(cr.ClownResamplerError) as e:
        p.low_resample_i32(st, padded, 100)
    assert e.value.code == cr.ERROR_NO_DEVICE and "no CPU fallback" in e.value.message
    assert st.astuple() == p.low_init(2, 44100, 48000, 44100)[1].astuple()      # state untouched
    with pytest.raises(cr.ClownResamplerError):
        p.low_resample_cb(st, padded, 100, lambda f: True)
    with pytest.raises(cr.ClownResamplerError):
        p.frame(st.raw.lowest_level, 2, padded, 0, 0)
    with pytest.raises(cr.ClownResamplerError):
        p.api.PlanCreate(st.raw, p.pre)
    ok, hs = p.high_init(2, 44100, 48000, 44100)
    with pytest.raises(cr.ClownResamplerError):
        p.high_run_i32(hs, ck.noise_pcm(200))


def test_highlevel_init_refuses_a_window_wider_than_the_staging_buffer():
    """The reference's HighLevel_Init accepts any radius; with 2 * radius * channels beyond its 0x1000-sample staging buffer its first
    refill asks the callback for a negative count of frames (clownresampler.h:1154) - undefined behaviour, found by tests/soak_gpu.py as
    heap corruption in THIS library, which sized its window by the same subtraction.  The product applies HighLevel_Adjust's rule
    (clownresampler.h:1202) at Init: refused, no error report, nothing allocated; the widest window that fits is still accepted."""
    p = _product.Product(8)
    for ch, rates in ((6, (43, 3, 1)), (13, (192000, 8000, 8000)), (16, (48000, 3000, 3000)), (1, (64000, 1, 1))):
        ok, _ = p.low_init(ch, *rates)
        if not ok:
            continue   # (refused by the low level already)
        st = p.api.LowLevel_State()
        p.api.LowLevel_Init(st, ch, *rates)
        radius = st.lowest_level.integer_stretched_kernel_radius
        assert radius * 2 >= 0x1000 // ch, (ch, rates, radius)
        ok, _ = p.high_init(ch, *rates)
        assert not ok and p.api.lib.ClownResamplerAMD_LastErrorCode() == 0, (ch, rates)
    # 2 channels 48 kHz -> 375 Hz at radius 8: 1024 frames of radius, 2 * 1024 * 2 = 0x1000 exactly: refused; one step milder: accepted
    assert not p.high_init(2, 48000, 375, 375)[0]
    ok, st = p.high_init(2, 48000, 376, 376)
    assert ok and st.max_radius_frames * 2 < 0x1000 // 2
    p.api.HighLevel_Release(st.raw)


def test_is_usable_answers_without_a_device(tmp_path):
    """ClownResamplerAMD_IsUsable: the question a drop-in client asks once at start-up - 0 here, where there is no GPU,
    without an error report and, in a plain C client with no handler installed, without a word on stderr - so that the
    client can choose the reference's own header instead of finding out in its first resample call."""
    p = _product.Product(3)
    assert p.api.IsUsable() == 0 and p.api.lib.ClownResamplerAMD_LastErrorCode() == 0
    src = tmp_path / "u.c"
    src.write_text('#include <stdio.h>\n#include "clownresampler_amd.h"\n'
                   'int main(void) { printf("%d %d\\n", ClownResamplerAMD_IsUsable(), ClownResamplerAMD_DeviceCount()); return 0; }\n')
    exe = tmp_path / "u"
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-Wall", "-Wno-long-long", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", os.path.dirname(cr.LIB_PATH),
                    "-lclownresampler_amd", "-Wl,-rpath," + os.path.dirname(cr.LIB_PATH)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["0", "0"], (r.returncode, r.stdout, r.stderr)


_NO_DEVICE_CLIENT = r'''
#include "clownresampler_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static ClownResampler_Precomputed p;
static ClownResampler_LowLevel_State s, before;
static ClownResampler_HighLevel_State h;
static size_t frames_seen = 0, pulled = 0;
static cc_bool take(void *u, const cc_s32f *f, cc_u8f n) { (void)u; (void)f; (void)n; ++frames_seen; return cc_true; }
static size_t pull(void *u, cc_s16l *b, size_t n) { (void)u; memset(b, 0, n * 2 * sizeof(*b)); pulled += n; return pulled > 100000 ? 0 : n; }
int main(void)
{
	static short in[4096 * 2];
	int out[64];
	size_t n = 4000;
	cc_bool ran_out = 7, r;
	if (ClownResamplerAMD_DeviceCount() > 0) return 77;
	ClownResampler_Precompute(&p);
	ClownResampler_LowLevel_Init(&s, 2, 44100, 48000, 44100);
	before = s;
	/* bulk: no frames, "stopped", state and count untouched, the reason on record */
	if (ClownResampler_LowLevel_ResampleBulk(&s, &p, in, &n, out, 8, &ran_out) != 0 || ran_out != cc_false || n != 4000 || memcmp(&s, &before, sizeof(s)) != 0) return 2;
	if (ClownResamplerAMD_LastErrorCode() != CLOWNRESAMPLER_AMD_ERROR_NO_DEVICE) return 3;
	/* callback form: cc_false = "the callback said stop" (clownresampler.h:746-748), nothing handed out, nothing consumed */
	ClownResamplerAMD_ClearError();
	r = ClownResampler_LowLevel_Resample(&s, &p, in, &n, take, NULL);
	if (r != cc_false || frames_seen != 0 || n != 4000 || memcmp(&s, &before, sizeof(s)) != 0 || ClownResamplerAMD_LastErrorCode() == 0) return 4;
	/* high level: control comes back (no endless loop), "the output callback said stop", no frames */
	ClownResamplerAMD_ClearError();
	if (!ClownResampler_HighLevel_Init(&h, 2, 44100, 48000, 44100)) return 5;
	r = ClownResampler_HighLevel_Resample(&h, &p, pull, take, NULL);
	if (r != cc_false || frames_seen != 0 || ClownResamplerAMD_LastErrorCode() == 0) return 6;
	puts("survived");
	return 0;
}
'''


def _no_device_client(tmp_path):
    src = tmp_path / "a.c"
    src.write_text(_NO_DEVICE_CLIENT)
    exe = tmp_path / "a"
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-Wall", "-Wno-long-long", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", os.path.dirname(cr.LIB_PATH),
                    "-lclownresampler_amd", "-Wl,-rpath," + os.path.dirname(cr.LIB_PATH)], check=True)
    return str(exe)


def test_default_error_handling_records_and_returns(tmp_path):
    """the reference has no failing path (clownresampler.h:746-748) and a drop-in must not end the host process over a
    failure it can hand back.  With NO handler installed every resample entry point returns the "callback said stop" outcome with the
    state and the input count untouched, ClownResamplerAMD_LastErrorCode says why, stderr carries the message - and the process lives."""
    r = subprocess.run([_no_device_client(tmp_path)], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "CLOWNRESAMPLER_AMD_ABORT_ON_ERROR"})
    if r.returncode == 77:
        pytest.skip("a GPU is present")
    assert r.returncode == 0 and r.stdout.strip() == "survived", (r.returncode, r.stdout, r.stderr)
    assert "clownresampler_amd: no usable HIP device" in r.stderr


def test_abort_on_error_is_opt_in(tmp_path):
    """CLOWNRESAMPLER_AMD_ABORT_ON_ERROR=1: the hard stop of rounds 1-5 - message, flight recorder, abort() - for whoever wants the core dump."""
    r = subprocess.run([_no_device_client(tmp_path)], capture_output=True, text=True, env=dict(os.environ, CLOWNRESAMPLER_AMD_ABORT_ON_ERROR="1"))
    if r.returncode == 77:
        pytest.skip("a GPU is present")
    assert r.returncode < 0 and "clownresampler_amd: no usable HIP device" in r.stderr and "flight recorder" in r.stderr


def test_device_code_keeps_its_promises(tmp_path):
    """Reads the gfx950 code objects out of the built library and checks two things the kernels' comments rely on and hipcc is
    free to break behind the source's back ("Tickets"):
    (1) no flat_ memory instruction anywhere: LDS words are addressed as LDS, global memory as global (a flat access counts in
        vmcnt AND lgkmcnt, and a volatile one is followed by vmcnt(0));
    (2) the SGPR of a ticket drawn with draw_ticket_begin (s_atomic_add, result in flight) is not touched by ANY instruction
        before the s_waitcnt lgkmcnt(0) of draw_ticket_end."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    lib = tmp_path / "lib.so"
    shutil.copy(cr.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True)
    bundles = sorted(p for p in tmp_path.iterdir() if "gfx950" in p.name)
    assert len(bundles) >= 10, [p.name for p in tmp_path.iterdir()]
    draws = pending_checked = 0
    for bundle in bundles:
        text = subprocess.run([objdump, "-d", str(bundle)], check=True, capture_output=True, text=True).stdout
        lines = [l.split("//")[0].strip() for l in text.splitlines()]
        lines = [l for l in lines if l and not l.endswith(":") and not l.startswith(("Disassembly", "/"))]
        flat = [l for l in lines if l.startswith("flat_")]
        assert not flat, (bundle.name, flat[:3])
        i = 0
        while i < len(lines):
            m = re.match(r"s_atomic_add s(\d+),", lines[i])
            if not m:
                i += 1
                continue
            draws += 1
            reg = int(m.group(1))
            j = i + 1
            while j < len(lines) and not re.match(r"s_waitcnt .*lgkmcnt\(0\)", lines[j]):
                ops = lines[j]
                singles = {int(x) for x in re.findall(r"\bs(\d+)\b", ops)}
                ranges = [(int(a), int(b)) for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", ops)]
                assert reg not in singles and not any(a <= reg <= b for a, b in ranges), (bundle.name, lines[i], ops)
                assert not ops.startswith(("s_endpgm", "s_atomic_add")), (bundle.name, lines[i], "no wait before", ops)
                j += 1
            assert j < len(lines), (bundle.name, lines[i], "never waited for")
            if j > i + 1:
                pending_checked += 1
            i = j
    assert draws >= 50 and pending_checked >= 2, (draws, pending_checked)   # every ticketed kernel; k_up2's two output forms


REFERENCE = "/root/reference"


def _symlink_tree(tmp_path, sub, names):
    """<tmp>/clownresampler.h -> include/clownresampler.h (+ the generated radius list it includes), <tmp>/<sub>/<name> -> the
    reference's own file.  Symlinks only: nothing of the reference is copied, nothing travels."""
    tree = tmp_path / "tree"
    (tree / sub).mkdir(parents=True)
    for h in os.listdir(os.path.join(ROOT, "include")):
        os.symlink(os.path.join(ROOT, "include", h), tree / h)
    for n in names:
        os.symlink(os.path.join(REFERENCE, sub, n), tree / sub / n)
    return tree


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="container only: needs the reference's own client sources")
@pytest.mark.parametrize("source", ["test-low-level.c", "test-high-level.c"])
def test_reference_test_harnesses_build_against_the_drop_in(tmp_path, source):
    """The source-level drop-in claim, pinned: the reference's UNMODIFIED harnesses (tests/test-low-level.c:25-28 and
    tests/test-high-level.c:25-27 define CLOWNRESAMPLER_IMPLEMENTATION + _STATIC and include "../clownresampler.h") compile as
    strict C89 against include/clownresampler.h and link against libclownresampler_amd.so the way tests/CMakeLists.txt:5-11 links
    them (plus the library) - and every ClownResampler_* function they call is then an UNDEFINED symbol the library provides, i.e.
    no reference implementation was compiled in."""
    tree = _symlink_tree(tmp_path, "tests", [source, "dr_flac.h"])
    exe = tmp_path / "harness"
    cmd = ["gcc", "-std=c89", "-pedantic", "-Wall", "-O1", str(tree / "tests" / source), "-o", str(exe),
           "-L", os.path.dirname(cr.LIB_PATH), "-Wl,-rpath," + os.path.dirname(cr.LIB_PATH), "-lclownresampler_amd", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "clownresampler.h" not in r.stderr, r.stderr[-3000:]          # not one diagnostic from our header under -pedantic
    undefined = set(re.findall(r"\bU (ClownResampler\w+)", subprocess.run(["nm", str(exe)], capture_output=True, text=True, check=True).stdout))
    expect = {"ClownResampler_Precompute", "ClownResampler_LowLevel_Init", "ClownResampler_LowLevel_Resample"} if source == "test-low-level.c" else \
             {"ClownResampler_Precompute", "ClownResampler_HighLevel_Init", "ClownResampler_HighLevel_Resample", "ClownResampler_HighLevel_ResampleEnd"}
    assert undefined == expect, undefined
    lib = C.CDLL(cr.LIB_PATH)
    assert all(hasattr(lib, n) for n in undefined)
    # and the usage message of the harness still comes out (no GPU is touched before the arguments are read)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode != 0 or out.stdout or out.stderr


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="container only: needs the reference's own client sources")
@pytest.mark.parametrize("source", ["low-level.c", "high-level.c"])
def test_reference_examples_compile_against_the_drop_in(tmp_path, source):
    """examples/low-level.c:42-45 / examples/high-level.c:42-44: the players include the header the same way.  They need an audio
    device and miniaudio's system libraries to LINK, so this is the compile half only (-fsyntax-only, strict C89 as the reference
    builds them) - enough to pin that every declaration, struct field and macro they use exists with a compatible type."""
    tree = _symlink_tree(tmp_path, "examples", [source, "libraries"])
    r = subprocess.run(["gcc", "-std=c89", "-pedantic", "-fsyntax-only", str(tree / "examples" / source)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "clownresampler.h" not in r.stderr, r.stderr[-3000:]


def test_dual_mono_guard_keeps_launches_inside_32_bit_descriptors():
    """the dual-mono kernels store both halves of a mono launch through ONE buffer descriptor of n_out * 4 bytes
    (clamped to 0xFFFFFFFC) with 32-bit byte offsets, and fetch both windows through one descriptor to the end of the caller's buffer:
    a launch whose output or input does not fit - with a tile of headroom for the ragged last tile's offsets - must stay on the mono
    kernels (64-bit pointers).  The predicate cr_plan_launch consults, on its own (host only)."""
    lib = C.CDLL(cr.LIB_PATH)
    f = lib.ClownResamplerAMD_DebugDualMonoFits
    f.restype, f.argtypes = C.c_int, [C.c_uint64] * 5
    inc = 60211                       # 44.1 -> 48 kHz
    tile = 4096
    ok = lambda n, in_bytes=1 << 30, t=tile, i=inc: f(n, (n + 1) // 2, t, i, in_bytes)
    assert ok(52_920_000) == 1                                  # ten minutes of mono: what the path is for
    assert ok((1 << 30) - 4 * tile - 1) == 1                    # the largest output that still leaves a tile of headroom ...
    assert ok((1 << 30) - 4 * tile) == 0                        # ... and the first that does not: n_out * 4 + 4 tiles * 4 > 0xFFFFFFFC
    assert ok(1 << 30) == 0 and ok((1 << 31) + 5) == 0          # 4 GiB of output and beyond: the second half would be silently dropped
    assert ok(52_920_000, in_bytes=0xFFFFFFFC) == 1 and ok(52_920_000, in_bytes=0xFFFFFFFD) == 0   # the input side: one descriptor over both windows
    assert f(1 << 31, 1 << 30, tile, inc, 1 << 30) == 0          # half itself
    assert f(100_000_000, 50_000_000, tile, 3000 << 16, 1 << 30) == 0   # the second window's distance (half * increment >> 16) beyond 2^31 frames
