// cr_instances.hpp - the instance tables of the polyphase kernels: which template instances exist, their tuning variants and
// geometries.  The instances are spread over several translation units (cr_inst_*.hip) so that they compile side by side and a
// change to one kernel rebuilds only the units that instantiate it; cr_kernels.hip (launch shim) merges what they provide.
#ifndef CR_INSTANCES_HPP
#define CR_INSTANCES_HPP

#include "cr_kpoly.hpp"
#include "cr_kwave.hpp"
#include "cr_kup.hpp"
#include "cr_kwave2.hpp"

namespace
{

// ---------------------------------------------------------------------------------------------------------
// Instance table of k_poly
// ---------------------------------------------------------------------------------------------------------
typedef void (*poly_fn)(const crhip_poly_launch);

// Tuning variants of the specialised instances: geometry x frames in flight x non-temporal stores.
// (A swizzled LDS row image, SWZ = 1, measured no better than the plain one and is not instantiated.)
//   variant = geo + 5 * ui + 10 * nt     geo: 0 (256 thr, 2 vec) 1 (512,1) 2 (512,2) 3 (1024,1) 4 (1024,2); ui: 0/1 -> U = 1/2
// All k_poly instances use the SDWA arithmetic (ASM = 1); the plain-C form (ASM = 0) is kept in the source as its
// readable definition, and k_generic is the independent 64-bit implementation the tests compare against the oracle too.
struct geometry
{
	int threads, vecs;
};
constexpr geometry GEOMETRY[5] = {{256, 2}, {512, 1}, {512, 2}, {1024, 1}, {1024, 2}};
constexpr int VARIANTS = 20;

template <int CH, int TT, int MODE, int NORM, int GEO, int ASM, int UI, int NT, int OUT16 = 0, int SWZ = 0>
constexpr poly_fn instance()
{
	return (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[GEO].threads, GEOMETRY[GEO].vecs, ASM, (1 << UI), SWZ, 0, OUT16, NT>;
}

template <int CH, int TT, int MODE, int NORM, int V>
struct variant_table
{
	static void fill(poly_fn *t)
	{
		t[V] = instance<CH, TT, MODE, NORM, V % 5, 1, (V / 5) % 2, (V / 10) % 2>();
		variant_table<CH, TT, MODE, NORM, V + 1>::fill(t);
	}
};
template <int CH, int TT, int MODE, int NORM>
struct variant_table<CH, TT, MODE, NORM, VARIANTS>
{
	static void fill(poly_fn *) {}
};

// specialised (channels, slots, mode, norm) instances; the BASELINE.json configurations
struct special
{
	uint32_t channels, slots, mode, norm;
	uint32_t default_variant;   // from tools/sweep_variants.py on MI355X (profiles/)
	poly_fn fn[VARIANTS];
	poly_fn fn16;               // int16-output form, default variant only
	poly_fn wave[2];            // k_wave (variants WAVE_VARIANT + {0: non-temporal stores, 1: plain}); nullptr if none
	poly_fn wave16;             // k_wave, int16 output
	bool dynamic_tiles;         // k_poly: draw tiles as tickets (measured per instance; see crhip_poly_launch.dynamic_tiles)
	poly_fn split[4];           // k_poly with two lanes per frame (variants SPLIT_VARIANT + i: geometry {4, 2} x nt {1, 0}); nullptr if none
	poly_fn up[2];              // k_up (variants UP_VARIANT + {0: 24-bit multiply-add + SDWA add per tap, 1: 64-bit multiply-add chain}); nullptr if none
	poly_fn up16;               // k_up, int16 output
	bool lite;                  // one k_poly instance only (the default variant, int32 and int16 forms): every variant id resolves to it
	uint32_t lite_lanes;        // lite instances: lanes per frame (2: each lane takes half of the channels, as the run-time instances above 8 channels do)
	uint32_t up_negmask;        // k_up / mad: bit s set = the weights of slot s are <= 0 in every row, clear = >= 0 (checked by the host per plan)
	poly_fn mad[2];             // the 64-bit multiply-add chain (compute_frame, ASM mode 2): variant MAD_VARIANT = k_poly geometry 3 with non-temporal stores,
	                            // MAD_VARIANT + 1 = k_wave where the instance has one, else k_poly geometry 3 with plain stores
	poly_fn wave2, wave2_16;    // k_wave2 (variant WAVE2_VARIANT): expanded window + 64-bit multiply-add taps; nullptr if none
	uint32_t wave2_waves, wave2_nvw, wave2_iter;   // its geometry (template WAVES, NVW, ITER)
	uint32_t wave2_fallback;    // the variant used instead when k_wave2 is the default and a plan does not qualify for it
	uint32_t wave2_fixed_signs; // 1: built for the slot signs in up_negmask (the host checks the plan's rows), 0: any rows
	uint32_t wave2_safemask;    // fixed signs: != 0 = the mov-armed form; these slots take weights up to 65536, the others only below it
	poly_fn mad16;              // the 64-bit chain with int16 output (non-temporal stores); nullptr: int16 output takes the SDWA form
	uint32_t lite_variant;      // lite instances with a chain: the variant id (geometry) of their SDWA form, which explicit k_poly variants resolve to
	uint32_t mad_frames;        // frames in flight per lane of the chain kernels (k_poly's U): 1 or 2
	poly_fn mad_rotated[2], mad16_rotated;   // the chain kernels with their rows rotated in LDS (SWZ): taken by launches whose plan asks for a rotation
	poly_fn fn_rotated, fn16_rotated;        // lite instances (pure upsampling, short windows): their one k_poly with rotated rows, likewise
	uint32_t mad_geo;                        // geometry (index into GEOMETRY) of the chain in mad[] where it is not the usual one (0: geometry 3 / the lite variant's)
	uint32_t mad_any_sign;                   // 1: the chain in mad[] is the any-sign form (ASM mode 3): no slot-sign precondition for the host to check
	poly_fn mad_dual, mad_dual_rotated;      // stereo instances: mad[0] built with DUAL (a mono stream as two phase-aligned "channels", crhip_poly_launch.dual); nullptr if none
	poly_fn wave2_dual;                      // ... and k_wave2 built with DUAL
	poly_fn mad_forms[6];                    // -DCRA_WITH_W2_FORMS only: timing-only forms of the rotated-rows chain kernel of the headline instance (see add_mad_forms)
	poly_fn wave2_forms[3];                  // -DCRA_WITH_W2_FORMS only: k_wave2's timing-only forms 1 ... 3 (ABL; results wrong); nullptr otherwise
};

constexpr uint32_t MAD_VARIANT = 28;    // variant ids 28, 29
constexpr uint32_t WAVE2_VARIANT = 30;  // variant id 30: k_wave2 where the instance has one
constexpr uint32_t RT_WAVE2S_VARIANT = 32;  // variant id 32: k_wave2s (wide frames, long windows; chosen by the host)
constexpr uint32_t RT_WAVE2_VARIANT = 31;   // variant id 31: the run-time-slot k_wave2 (plans without a specialised instance; chosen by the host)

// k_wave2 of an instance: fixed slot signs (NEGMASK != 0: pure upsampling, 2 VALU per tap and channel) or any rows (3)
// SAFEMASK (fixed signs only): see k_wave2 - 15 slots of the 8-lobe table: slots 7 and 8 are the ones that reach 65536
template <int CH, int TT, int MODE, int NORM, int WAVES, int NVW, int ITER, unsigned NEGMASK>
void add_wave2(special &s)
{
	constexpr int SIGNED = NEGMASK == 0 ? 1 : 0;
	constexpr unsigned SAFEMASK = (!SIGNED && TT == 15) ? 0x180u : 0u;
	s.wave2 = (poly_fn)k_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, 0, 1, NEGMASK, SIGNED, SAFEMASK>;
	s.wave2_16 = (poly_fn)k_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, 1, 1, NEGMASK, SIGNED, SAFEMASK>;
	if constexpr (CH == 2)
		s.wave2_dual = (poly_fn)k_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, 0, 1, NEGMASK, SIGNED, SAFEMASK, 1>;   // a mono stream as two phase-aligned channels
#ifdef CRA_WITH_W2_FORMS
	if constexpr (CH == 2 && (TT == 15 || TT == 17 || TT == 33))
	{
		s.wave2_forms[0] = (poly_fn)k_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, 0, 1, NEGMASK, SIGNED, SAFEMASK, 0, 1>;
		s.wave2_forms[1] = (poly_fn)k_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, 0, 1, NEGMASK, SIGNED, SAFEMASK, 0, 2>;
		s.wave2_forms[2] = (poly_fn)k_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, 0, 1, NEGMASK, SIGNED, SAFEMASK, 0, 3>;
	}
#endif
	s.wave2_safemask = SAFEMASK;
	s.wave2_waves = WAVES;
	s.wave2_nvw = NVW;
	s.wave2_iter = ITER;
	s.wave2_fixed_signs = SIGNED ? 0u : 1u;
	if (!SIGNED)
		s.up_negmask = NEGMASK;
}

// The 64-bit-chain instances of k_poly come in two forms: rows plain in LDS, and rows rotated within their blocks of 16 (SWZ; the
// plan picks the rotation for its increment).  At exactly 8x / 16x upsampling the rows of 16 neighbouring lanes are 128 / 64 apart
// and share an LDS bank slot: rotated, mono 16x takes 74 us instead of 210, 8x 74 instead of 116, 6 channels 16x 103 instead of
// 119 - but the three instructions per frame cost the ratios without conflicts 1-4 % (profiles/r02_chain_rotated_rows.log), so a
// plan asks for a rotation only where the conflict model (cr_poly_pick_swizzle) says it pays.

// The 64-bit chain (variants 28 / 29, + int16 form) added to a full instance that has no k_up form - mono - and made its default
template <int CH, int TT, int MODE, int NORM, unsigned UPMASK, int U = 1>
special with_chain(special s)
{
	s.mad[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), U, 0, 0, 0, 1>;
	s.mad_rotated[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), U, 1, 0, 0, 1>;
	s.mad[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), U, 0, 0, 0, 0>;
	s.mad_rotated[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), U, 1, 0, 0, 0>;
	s.mad16 = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), U, 0, 0, 1, 1>;
	s.mad16_rotated = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), U, 1, 0, 1, 1>;
	s.up_negmask = UPMASK;
	s.mad_frames = U;
	return s;
}

// The ANY-SIGN 64-bit chain (ASM mode 3: 3 instructions per tap and channel instead of the SDWA form's 4) behind variants 28 / 29 of a
// downsampling instance with an even channel count, at geometry GEO; DEFAULT makes it the instance's default
template <int CH, int TT, int MODE, int NORM, int GEO, bool DEFAULT = false, int U = 1>   // U: frames in flight per lane
special with_signed_chain(special s)
{
	constexpr int T = GEOMETRY[GEO].threads, V = GEOMETRY[GEO].vecs;
	s.mad[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, T, V, 3, U, 0, 0, 0, 1>;
	s.mad[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, T, V, 3, U, 0, 0, 0, 0>;
	s.mad16 = (poly_fn)k_poly<CH, TT, MODE, NORM, T, V, 3, U, 0, 0, 1, 1>;
	if constexpr (CH == 2 && TT <= 8 && U == 1)
		s.mad_dual = (poly_fn)k_poly<CH, TT, MODE, NORM, T, V, 3, U, 0, 0, 0, 1, 1, 0, 1>;
	s.mad_any_sign = 1u;
	s.mad_frames = U;
	s.mad_geo = GEO;
	if (s.lite)
	{
		// (a lite instance with a chain: the chain IS its default - resolve_variant - and every k_poly variant id its SDWA form)
		s.lite_variant = s.default_variant;
		s.default_variant = MAD_VARIANT;
	}
	else if (DEFAULT)
		s.default_variant = MAD_VARIANT;
	return s;
}

// DEFAULT: k_wave2 becomes the instance's default variant (where it measured faster than the k_poly / k_wave forms); the
// previous default stays the fallback for plans whose window does not fit a wave's slice
template <int CH, int TT, int MODE, int NORM, int WAVES, int NVW, int ITER, unsigned NEGMASK, bool DEFAULT = false>
special with_wave2(special s)
{
	add_wave2<CH, TT, MODE, NORM, WAVES, NVW, ITER, NEGMASK>(s);
	if (DEFAULT)
	{
		s.wave2_fallback = s.default_variant;
		s.default_variant = WAVE2_VARIANT;
	}
	return s;
}

constexpr uint32_t UP_VARIANT = 26;     // variant ids 26, 27 select k_up where the instance has one and the plan qualifies
constexpr int UP_WAVES = 12;          // (16 - four waves per SIMD, which k_up2's 128 VGPRs allow - leaves each wave 3.5 KB of staging beside the 66 KB of rows: wave-tiles of 36 positions
                                      //  instead of 60, cfg 3 179 against 132 us: profiles/r04_kup2_ab.log)
constexpr uint32_t UP_MAX_WAVE_TILE = 1024;   // output frames per wave-tile (LDS staging)

constexpr uint32_t SPLIT_VARIANT = 22;   // variant ids 22..25
constexpr int SPLIT_GEO[4] = {4, 2, 4, 2};
constexpr int SPLIT_NT[4] = {1, 1, 0, 0};

constexpr uint32_t WAVE_VARIANT = 20;   // variant ids 20, 21 select k_wave where the instance has one
constexpr int WAVE_WAVES = 16, WAVE_NVW = 1, WAVE_ITER = 4;

template <int CH, int TT, int MODE, int NORM, int DV, bool WAVE = false, bool DYNAMIC = false, unsigned UPMASK = 0>
special make_special()
{
	special s = {CH, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, DYNAMIC, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, false, 1u, UPMASK, {nullptr, nullptr}, nullptr, nullptr, 0u, 0u, 0u, 13u, 0u, 0u, nullptr, 13u, 1u};
	if constexpr (UPMASK != 0 && CH % 2 == 0)
	{
		s.mad[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 1>;
		s.mad_rotated[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 0, 0, 1>;
#ifdef CRA_WITH_W2_FORMS
		if constexpr (CH == 2 && TT == 5)
		{
			// k_poly's ABL: + 16 window reads without conflicts, + 32 row reads without; low codes 1 = no stores, 3 = no stores and no DMA
			s.mad_forms[0] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 16, 0, 1>;
			s.mad_forms[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 32, 0, 1>;
			s.mad_forms[2] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 48, 0, 1>;
			s.mad_forms[3] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 64 + 3, 0, 1>;
			s.mad_forms[4] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 48 + 3, 0, 1>;
			s.mad_forms[5] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 64 + 1, 0, 1>;
		}
#endif
		s.mad[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 0>;
		s.mad_rotated[1] = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 0, 0, 0>;
		s.mad16 = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 0, 0, 1, 1>;
		s.mad16_rotated = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 0, 1, 1>;
		if constexpr (CH == 2 && TT <= 8)
		{
			s.mad_dual = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 1, 1, 0, 1>;
			s.mad_dual_rotated = (poly_fn)k_poly<CH, TT, MODE, NORM, GEOMETRY[3].threads, GEOMETRY[3].vecs, (int)(2u | (UPMASK << 8)), 1, 1, 0, 0, 1, 1, 0, 1>;
		}
	}
	if constexpr (UPMASK != 0)
	{
		static_assert(MODE == CRHIP_ROWMODE_UPSAMPLE, "k_up is for pure upsampling");
		s.up[0] = (poly_fn)k_up<CH, TT, NORM, UPMASK, UP_WAVES, 0, 1, 0, 1>;   // round 1's form (64-bit chain, bias registers): kept for comparison
		if constexpr (CH == 2)
		{
			// stereo: variant 26 = k_up2 with the integer chain (v_mov_b32 + v_mad_i64_i32 per tap), variant 27 (the default where k_up
			// is one) = k_up2 with the FP32 round-toward-zero chain (one v_pk_fma_f32 per tap)
			s.up[0] = (poly_fn)k_up2<CH, TT, NORM, UPMASK, UP_WAVES, 0, 1, 0, 0>;
			s.up[1] = (poly_fn)k_up2<CH, TT, NORM, UPMASK, UP_WAVES, 0, 1, 0, 1>;
			s.up16 = (poly_fn)k_up2<CH, TT, NORM, UPMASK, UP_WAVES, 1, 1, 0, 1>;
		}
		else
		{
			s.up[1] = s.up[0];
			s.up16 = (poly_fn)k_up<CH, TT, NORM, UPMASK, UP_WAVES, 1, 1, 0, 1>;
		}
	}
	variant_table<CH, TT, MODE, NORM, 0>::fill(s.fn);
	constexpr int KV = DV < 20 ? DV : 13;   // the k_poly variant behind a k_wave default (its fallback and int16 geometry)
	s.fn16 = instance<CH, TT, MODE, NORM, KV % 5, 1, (KV / 5) % 2, (KV / 10) % 2, 1>();
	if constexpr (CH % 2 == 0 && CH >= 8)
	{
		// two lanes per frame, each with CH / 2 channels
		s.split[0] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[4].threads, GEOMETRY[4].vecs, 1, 1, 0, 0, 0, 1, 2>;
		s.split[1] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[2].threads, GEOMETRY[2].vecs, 1, 1, 0, 0, 0, 1, 2>;
		s.split[2] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[4].threads, GEOMETRY[4].vecs, 1, 1, 0, 0, 0, 0, 2>;
		s.split[3] = (poly_fn)k_poly<CH / 2, TT, MODE, NORM, GEOMETRY[2].threads, GEOMETRY[2].vecs, 1, 1, 0, 0, 0, 0, 2>;
	}
	if constexpr (WAVE)
	{
		s.wave[0] = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 1>;
		s.wave[1] = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 0, 0>;
		s.wave16 = (poly_fn)k_wave<CH, TT, MODE, NORM, WAVE_WAVES, WAVE_NVW, WAVE_ITER, 1, 1>;
	}
	return s;
}

// A specialised instance WITHOUT the tuning variants: one k_poly (compile-time slot count, pipelined LDS reads) at the
// geometry the run-time-slot instance of that channel count uses, non-temporal stores, int32 and int16 forms.  For the common
// surround layouts, where the run-time-slot loop leaves 10-20 % behind (profiles/r01_channel_table.log).
template <int CH, int TT, int MODE, int NORM, int DV = (CH <= 4 ? 13 : 14)>   // default: (1024 threads, 1 or 2 vectors per thread), one frame in flight, non-temporal stores
special make_special_lite()
{
	special s = {CH, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, false, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, true, 1u, 0u, {nullptr, nullptr}, nullptr, nullptr, 0u, 0u, 0u, 13u, 0u, 0u, nullptr, 13u, 1u};
	const poly_fn fn = instance<CH, TT, MODE, NORM, DV % 5, 1, (DV / 5) % 2, (DV / 10) % 2>();
	for (int v = 0; v < VARIANTS; ++v)
		s.fn[v] = fn;
	s.fn16 = instance<CH, TT, MODE, NORM, DV % 5, 1, (DV / 5) % 2, (DV / 10) % 2, 1>();
	if constexpr (MODE == CRHIP_ROWMODE_UPSAMPLE && TT <= 8)
	{
		// (where the rows of neighbouring lanes collide in LDS - exactly 8x / 16x upsampling - see CHAIN instances above)
		s.fn_rotated = instance<CH, TT, MODE, NORM, DV % 5, 1, (DV / 5) % 2, (DV / 10) % 2, 0, 1>();
		s.fn16_rotated = instance<CH, TT, MODE, NORM, DV % 5, 1, (DV / 5) % 2, (DV / 10) % 2, 1, 1>();
	}
	return s;
}

// ... whose default is the 64-bit chain (pure upsampling, even channel counts: UPMASK = the slots with negative weights), at the same
// geometry; the SDWA form stays behind every explicit k_poly variant id - the fallback for plans whose rows do not match the masks
template <int CH, int TT, int NORM, unsigned UPMASK, int DV = (CH <= 4 ? 13 : 14)>
special make_special_lite_chain()
{
	static_assert(CH % 2 == 0 && UPMASK != 0, "the chain works on packed pairs of channels, with fixed slot signs");
	special s = make_special_lite<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, DV>();
	constexpr int T = GEOMETRY[DV % 5].threads, V = GEOMETRY[DV % 5].vecs;
	s.mad[0] = (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, T, V, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 1>;
	s.mad_rotated[0] = (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, T, V, (int)(2u | (UPMASK << 8)), 1, 1, 0, 0, 1>;
	s.mad[1] = (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, T, V, (int)(2u | (UPMASK << 8)), 1, 0, 0, 0, 0>;
	s.mad_rotated[1] = (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, T, V, (int)(2u | (UPMASK << 8)), 1, 1, 0, 0, 0>;
	s.mad16 = (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, T, V, (int)(2u | (UPMASK << 8)), 1, 0, 0, 1, 1>;
	s.mad16_rotated = (poly_fn)k_poly<CH, TT, CRHIP_ROWMODE_UPSAMPLE, NORM, T, V, (int)(2u | (UPMASK << 8)), 1, 1, 0, 1, 1>;
	s.up_negmask = UPMASK;
	s.lite_variant = DV;
	s.default_variant = MAD_VARIANT;
	return s;
}

// ... and with two lanes per frame (CHT channels in all, CHT / 2 per lane), at the geometry of the run-time instances above 8 channels
// Tap arithmetic: 3 = the any-sign 64-bit chain (3 VALU per tap and channel), 1 = the SDWA form (4).  One box, 9 to 16 channels at
// 44.1 <-> 48 kHz: the chain 1-8 % faster on every row but one (profiles/r04_odd_wide_frames_ab.log).
#ifndef CRA_SPLIT_LITE_ASM
 #define CRA_SPLIT_LITE_ASM 3
#endif
template <int CHT, int TT, int MODE, int NORM>
special make_special_lite_split()
{
	constexpr int DV = 14;   // (1024 threads, 2 vectors per thread), one frame in flight, non-temporal stores
	constexpr int T = GEOMETRY[DV % 5].threads, V = GEOMETRY[DV % 5].vecs;
	special s = {CHT, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, false, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, true, 2u, 0u, {nullptr, nullptr}, nullptr, nullptr, 0u, 0u, 0u, 13u, 0u, 0u, nullptr, 13u, 1u};
	const poly_fn fn = (poly_fn)k_poly<CHT / 2, TT, MODE, NORM, T, V, CRA_SPLIT_LITE_ASM, 1, 0, 0, 0, 1, 2>;
	for (int v = 0; v < VARIANTS; ++v)
		s.fn[v] = fn;
	s.fn16 = (poly_fn)k_poly<CHT / 2, TT, MODE, NORM, T, V, CRA_SPLIT_LITE_ASM, 1, 0, 0, 1, 1, 2>;
	return s;
}

// ... the same for an ODD total above 8: the second lane's last channel is k_poly's phantom (PH)
template <int CHT, int TT, int MODE, int NORM>
special make_special_lite_split_odd()
{
	static_assert(CHT % 2 == 1 && CHT > 8, "odd channel counts above 8");
	constexpr int DV = 14;
	constexpr int T = GEOMETRY[DV % 5].threads, V = GEOMETRY[DV % 5].vecs;
	special s = {CHT, TT, MODE, NORM, DV, {}, nullptr, {nullptr, nullptr}, nullptr, false, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr}, nullptr, true, 2u, 0u, {nullptr, nullptr}, nullptr, nullptr, 0u, 0u, 0u, 13u, 0u, 0u, nullptr, 13u, 1u};
	const poly_fn fn = (poly_fn)k_poly<(CHT + 1) / 2, TT, MODE, NORM, T, V, CRA_SPLIT_LITE_ASM, 1, 0, 0, 0, 1, 2, 1>;
	for (int v = 0; v < VARIANTS; ++v)
		s.fn[v] = fn;
	s.fn16 = (poly_fn)k_poly<(CHT + 1) / 2, TT, MODE, NORM, T, V, CRA_SPLIT_LITE_ASM, 1, 0, 0, 1, 1, 2, 1>;
	return s;
}

// run-time slot count: every channel count 1..8 and the even counts 10..16 (the reference's maximum,
// CLOWNRESAMPLER_MAXIMUM_CHANNELS, clownresampler.h:462), both row modes, both normalisations, both output forms.  One
// geometry per channel count - 1024 threads; 16 KiB tiles for up to 4 channels, 32 KiB above (an 8-channel frame is 16
// bytes) - SDWA arithmetic, one frame in flight, non-temporal stores.  Above 8 channels a frame is shared by TWO neighbouring
// lanes (k_poly's SPLIT), each taking half of its channels: the per-lane code is that of 5..8 channels.
constexpr int runtime_geo(int channels)
{
	return channels <= 4 ? 3 : 4;   // (512 threads x 2 vectors, which the specialised 8-channel instances prefer, measured 10-20 % slower here)
}
// 1 = the run-time-slot instances stage their rows rotated (the plan picks the rotation for its increment).  Measured with 1 on
// every channel count and ratio of tools/channel_table.py: no gain anywhere, 0-5 % slower (profiles/
// r02_channel_table_runtime_rows_swizzled.log against r02_channel_table.log) - bank conflicts are half of these kernels' LDS
// cycles, but the LDS is not what binds them - so it stays 0; k_wave2, which IS LDS-bound, rotates its rows.
constexpr int RUNTIME_SWZ = 0;
// Tap arithmetic of the run-time-slot instances: 3 = one any-sign 64-bit multiply-add chain per channel (3 VALU per tap and channel:
// shift or mask, SDWA xor to arm, v_mad_i64_i32), 1 = the SDWA form (4).  Mono keeps its packed-window SDWA taps either way.
#ifndef CRA_RUNTIME_ASM
 #define CRA_RUNTIME_ASM 3
#endif
constexpr int RUNTIME_ASM = CRA_RUNTIME_ASM;
// bytes a frame of the run-time-slot instance of CHT channels takes in the LDS tiles it computes from, where k_poly repacks them (0: as they come)
template <int CHT>
constexpr uint32_t runtime_padded_bytes()
{
	return padded_frames<(CHT + 1) / 2, 0, 2, CHT % 2>() ? 32u : 0u;
}
constexpr int runtime_split(int channels)
{
	return channels > 8 ? 2 : 1;
}

template <int CH, int OUT16>
poly_fn pick_runtime(uint32_t mode, uint32_t norm)
{
	constexpr int GEO = runtime_geo(CH);
	if (norm == CRHIP_NORM_S31)
		return mode == CRHIP_ROWMODE_UPSAMPLE ? instance<CH, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, GEO, RUNTIME_ASM, 0, 1, OUT16, RUNTIME_SWZ>()
		                                      : instance<CH, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, GEO, RUNTIME_ASM, 0, 1, OUT16, RUNTIME_SWZ>();
	return mode == CRHIP_ROWMODE_UPSAMPLE ? instance<CH, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, GEO, RUNTIME_ASM, 0, 1, OUT16, RUNTIME_SWZ>()
	                                      : instance<CH, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_U32, GEO, RUNTIME_ASM, 0, 1, OUT16, RUNTIME_SWZ>();
}

// two lanes per frame, HALF channels each (run-time slot count, geometry 4); PH = 1: 2 * HALF - 1 channels (see k_poly)
template <int HALF, int OUT16, int PH = 0, int PADT = 0>
poly_fn pick_runtime_split(uint32_t mode, uint32_t norm, uint32_t padded = 0)
{
	constexpr int T = GEOMETRY[runtime_geo(16)].threads, V = GEOMETRY[runtime_geo(16)].vecs;
	if constexpr (PADT == 0 && padded_frames<HALF, 0, 2, PH>())
	{
		if (padded)
			return pick_runtime_split<HALF, OUT16, PH, 1>(mode, norm);   // the same instance computing from padded tiles (k_poly PADT)
	}
	if (norm == CRHIP_NORM_S31)
		return mode == CRHIP_ROWMODE_UPSAMPLE ? (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_S31, T, V, RUNTIME_ASM, 1, RUNTIME_SWZ, 0, OUT16, 1, 2, PH, 0, PADT>
		                                      : (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, T, V, RUNTIME_ASM, 1, RUNTIME_SWZ, 0, OUT16, 1, 2, PH, 0, PADT>;
	return mode == CRHIP_ROWMODE_UPSAMPLE ? (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_UPSAMPLE, CRHIP_NORM_U32, T, V, RUNTIME_ASM, 1, RUNTIME_SWZ, 0, OUT16, 1, 2, PH, 0, PADT>
	                                      : (poly_fn)k_poly<HALF, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_U32, T, V, RUNTIME_ASM, 1, RUNTIME_SWZ, 0, OUT16, 1, 2, PH, 0, PADT>;
}


} // namespace

// what the instance units provide (external linkage; `special` is declared identically in every unit)
namespace crk
{
int specials_headline(void *table, int capacity);   // cr_inst_headline.hip: stereo / mono 3-lobe instances of BASELINE configs[1] / [4] + ablations
int specials_long(void *table, int capacity);       // cr_inst_long.hip: the 8-lobe (15- and 17-slot) instances for mono and stereo, BASELINE configs[2]
int specials_long_b(void *table, int capacity);     // cr_inst_long_b.hip: the same for 3 to 6 channels
int specials_multi_a(void *table, int capacity);    // cr_inst_multi_a.hip: 8 channels 48 -> 44.1 kHz with every tuning variant, BASELINE configs[3]
int specials_multi_b(void *table, int capacity);    // cr_inst_multi_b.hip: 3 to 16 channels at 44.1 <-> 48 kHz, one instance each
int specials_multi_c(void *table, int capacity);    // cr_inst_multi_c.hip: the odd channel counts above 8 at 44.1 <-> 48 kHz
int specials_down(void *table, int capacity);       // cr_inst_down.hip: mono / stereo at the usual downsampling ratios
void *ablation_instance(int abl);                   // cr_inst_headline.hip / cr_inst_long.hip (abl 8)
void *ablation_instance_long(int abl);
void *runtime_instance_1_4(uint32_t channels, uint32_t mode, uint32_t norm, int out16);     // cr_inst_runtime_a.hip
void *runtime_instance_5_8(uint32_t channels, uint32_t mode, uint32_t norm, int out16);     // cr_inst_runtime_b.hip
void *runtime_instance_9_16(uint32_t channels, uint32_t mode, uint32_t norm, int out16, uint32_t padded);    // cr_inst_runtime_c.hip (padded: k_poly's PADT form, where the channel count has one)
void *runtime_wave2s_instance(uint32_t channels, int out16);                                   // cr_inst_runtime_w.hip: k_wave2s, a lane per channel pair
void *runtime_wave2_instance(uint32_t channels, uint32_t mode, int out16);                  // cr_inst_runtime_w.hip: k_wave2, run-time slot count
}

#endif // CR_INSTANCES_HPP
