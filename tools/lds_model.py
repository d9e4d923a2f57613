#!/usr/bin/env python3
"""What the host's LDS bank-conflict models say about a ratio (no GPU needed: cr_plan.c through ctypes).
usage: lds_model.py <radius> <channels> <in_rate>:<out_rate> ..."""
import ctypes as C, sys, os, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import clownresampler_amd as cr


class Cfg(C.Structure):
    _fields_ = [("skr", C.c_uint64), ("radius_frames", C.c_uint64), ("delta", C.c_uint64), ("step", C.c_uint64)]


class Poly(C.Structure):
    _fields_ = [("eligible", C.c_int), ("reason", C.c_char_p), ("fatal", C.c_int)] + [(n, C.c_uint32) for n in (
        "slots", "first_slot", "shifted", "first_mr", "window_extra", "rel_first", "rows", "row_stride", "row_mode")] + [
        ("aff_a", C.c_int32), ("aff_b", C.c_int32), ("aff_c", C.c_int32), ("delta", C.c_uint32), ("skr", C.c_uint32), ("step", C.c_uint32),
        ("norm_mode", C.c_uint32), ("weights", C.POINTER(C.c_int32))]


def main():
    radius, channels = int(sys.argv[1]), int(sys.argv[2])
    api = cr.load(radius)
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "bin", "libcrplan.so")   # cr_plan.c alone: the product library exports only its API
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(here, "..", "clownresampler_amd", "csrc", "cr_plan.c"), "-lm"], check=True)
    lib = C.CDLL(so)
    pre = api.Precomputed()
    api.Precompute(pre)
    table = (C.c_int32 * api.table_len)(*[int(v) for v in pre.lanczos_kernel_table])
    lib.cr_window_conflicts.restype = C.c_double
    for spec in sys.argv[3:]:
        a, b = (int(v) for v in spec.split(":"))
        st = api.LowLevel_State()
        api.LowLevel_Init(st, channels, a, b, min(a, b))
        ll = st.lowest_level
        cfg = Cfg(ll.stretched_kernel_radius, ll.integer_stretched_kernel_radius, ll.stretched_kernel_radius_delta, ll.kernel_step_size)
        poly = Poly()
        lib.cr_poly_build(table, C.c_size_t(api.table_len), C.byref(cfg), C.byref(poly))
        inc = (st.increment if hasattr(st, "increment") else None)
        inc = int(st.increment)
        print("%s radius %d ch %d: increment %d (%.5f), slots %d rows %d stride %d shifted %d window_extra %d; rows image %d bytes" % (
            spec, radius, channels, inc, inc / 65536.0, poly.slots, poly.rows, poly.row_stride, poly.shifted, poly.window_extra,
            ((poly.rows + 15) // 16 * 16) * poly.row_stride * 4))
        for lm in (0, 1):
            plain, best = C.c_double(), C.c_double()
            lib.cr_poly_pick_swizzle_mapped.restype = C.c_uint32
            k = lib.cr_poly_pick_swizzle_mapped(C.byref(poly), C.c_uint64(inc), C.c_uint32(lm), C.byref(plain), C.byref(best))
            w = lib.cr_window_conflicts(C.byref(poly), C.byref(cfg), C.c_uint64(inc), C.c_uint32(channels), C.c_uint32(lm))
            print("   lane_map %d: row reads (b128, 4 cycles): +%.2f plain, +%.2f with rotation %d; window reads (2 cycles): +%.2f" % (lm, plain.value, best.value, k, w))


main()
