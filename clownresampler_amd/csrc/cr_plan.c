/*
 * cr_plan.c - host-side planning for the GPU hot path.  Pure C, no HIP.
 *
 * 1. Closed forms of the reference's output-timeline walk (clownresampler.h:1058-1092): how many frames a call
 *    emits, where the position is after n frames, how much input n frames can touch.  The GPU computes frame j
 *    directly at position P0 + j * increment, so the host needs these to size launches and to leave the caller's
 *    state exactly as the reference's frame-by-frame loop would.
 *
 * 2. Polyphase rows.  For a fixed configuration the taps of an output frame depend only on its 16-bit fractional
 *    position: the reference derives (min_relative, max_relative, kernel_start) from it (clownresampler.h:993-1001)
 *    and then walks the Lanczos table with stride kernel_step_size (:1008).  cr_poly_build() performs that walk
 *    once per distinct (min_relative, max_relative, kernel_start) on the host and stores, per such "phase row",
 *    the weights laid out on a COMMON window of `slots` consecutive input frames starting `first_slot` frames
 *    after position_integer (zero where the phase has no tap: a zero weight contributes exactly 0 to every
 *    truncated product and to the weight sum), followed by the exact 17.15 reciprocal 0x80000000 / sum (:1025).
 *    Window columns that are zero in every row are trimmed (e.g. for pure upsampling the reference's sixth tap
 *    only exists at fraction 0 and lands on table[0] == 0, so 5 slots suffice).
 *    The device then needs one row index per frame instead of a strided table walk and an integer divide.
 *
 *    Row index, two forms (the device mirrors are in cr_kernels.hip row_of<>):
 *      UPSAMPLE  delta == 0 && step == 1024:  row = (65536 - frac) >> 6          (1025 rows)
 *      AFFINE    otherwise:                    row = kernel_start + A*min_relative + B*max_relative + C
 *                min_relative and max_relative each take at most two consecutive values and change at most once
 *                as frac grows, so at most three (min,max) combinations occur and an affine function of the two
 *                can give each combination its own contiguous block of rows.
 *    Whatever the form, the builder walks ALL 65536 fractions and checks that every one maps to a row whose
 *    contents equal what the reference's definition gives for that fraction.
 *
 *    The fast kernel is 32-bit; `eligible` records whether its preconditions hold for this table/configuration:
 *    |weight| < 2^23, 0 < reciprocal < 2^23, |accumulator| < 2^23 and |accumulator * reciprocal| < 2^31 (signed
 *    normalisation) or < 2^32 (magnitude normalisation, norm_mode), with |accumulator| bounded by
 *    sum(|weight|)/2 + slots (|sample| <= 32768, each term truncated).
 */
#include "cr_plan.h"

#include <stdlib.h>
#include <string.h>

#include "crhip.h"

#define FRAC_ONE 65536u

void cr_phase_of(const cr_config *cfg, uint32_t frac, cr_phase *out)
{
	/* reference clownresampler.h:993-994, :1001, :1008 (tap count = number of loop iterations) */
	const uint64_t first_rel = ((uint64_t)frac + cfg->delta + (FRAC_ONE - 1)) / FRAC_ONE;
	const uint64_t last_rel = ((uint64_t)frac + cfg->skr) / FRAC_ONE;

	out->first_rel = (uint32_t)first_rel;
	out->last_rel = (uint32_t)last_rel;
	out->taps = (uint32_t)(cfg->radius_frames + last_rel - first_rel);
	out->table_at = cfg->step * (first_rel * FRAC_ONE - frac) / FRAC_ONE;
}

uint32_t cr_poly_row_of(const cr_poly *poly, uint32_t frac)
{
	if (poly->row_mode == CRHIP_ROWMODE_UPSAMPLE)
	{
		return (FRAC_ONE - frac) >> 6;
	}
	else
	{
		/* must stay in step with row_of<CRHIP_ROWMODE_AFFINE> in cr_kernels.hip (32-bit arithmetic) */
		const uint32_t mr = (frac + poly->delta + (FRAC_ONE - 1)) >> 16;
		const uint32_t xr = (frac + poly->skr) >> 16;
		const uint32_t kstart = (poly->step * ((mr << 16) - frac)) >> 16;
		return (uint32_t)((int32_t)kstart + poly->aff_a * (int32_t)mr + poly->aff_b * (int32_t)xr + poly->aff_c);
	}
}

int cr_poly_periodic(const cr_poly *poly, uint64_t increment, uint32_t frac, uint32_t period, uint32_t *rows, uint32_t *starts)
{
	uint32_t p;

	for (p = 0; p < period; ++p)
	{
		const uint64_t pos = (uint64_t)frac + (uint64_t)p * increment;
		const uint32_t f = (uint32_t)(pos & (FRAC_ONE - 1));
		const uint32_t mr = (f + poly->delta + (FRAC_ONE - 1)) >> 16;

		rows[p] = cr_poly_row_of(poly, f);
		if (rows[p] >= poly->rows)
			return 0;
		starts[p] = (uint32_t)(pos >> 16) + poly->first_slot + (poly->shifted ? mr - poly->first_mr : 0u);
	}
	return 1;
}

void cr_poly_free(cr_poly *poly)
{
	free(poly->weights);
	poly->weights = NULL;
}

static int fail(cr_poly *out, int fatal, const char *reason)
{
	out->eligible = 0;
	out->fatal = fatal;
	out->reason = reason;
	return fatal;
}

typedef struct combo
{
	uint32_t mr, xr;
	uint64_t klo, khi;
	uint32_t base; /* first row of this combination's block */
} combo;

int cr_poly_build(const int32_t *table, size_t table_len, const cr_config *cfg, cr_poly *out)
{
	combo combos[4];
	unsigned ncombos = 0;
	uint32_t frac;
	uint32_t nz_lo = 0xFFFFFFFFu, nz_hi = 0; /* frame-offset range [nz_lo, nz_hi) holding non-zero weights */
	uint32_t off_lo = 0xFFFFFFFFu, off_hi = 0; /* frame-offset range holding taps at all */
	uint32_t rel_lo = 0xFFFFFFFFu, rel_hi = 0; /* the same non-zero range counted from each phase's OWN first tap (min_relative) */
	uint32_t mr_lo = 0xFFFFFFFFu, mr_hi = 0;   /* range of min_relative */
	int64_t max_weight = 0, min_weight = 0;   /* over every tap any phase uses */

	memset(out, 0, sizeof(*out));
	out->reason = "";

	if (cfg->delta >= FRAC_ONE || cfg->skr >= ((uint64_t)1 << 40) || cfg->radius_frames >= ((uint64_t)1 << 24) || cfg->step > 0xFFFFFFFFu)
		return fail(out, 1, "configuration scalars are not what ClownResampler_LowestLevel_Configure produces");

	/* ---- pass 1: every fraction's tap window; table bounds; row blocks; non-zero window ----
	   Neighbouring fractions mostly share (min_relative, max_relative, kernel_start); the table walk is only
	   repeated when that key changes, so the cost is about rows * taps ~ the table size, whatever the stretch. */
	{
		cr_phase prev;
		int have_prev = 0;

		for (frac = 0; frac < FRAC_ONE; ++frac)
		{
			cr_phase ph;
			uint32_t t;
			unsigned c;
			int64_t sum = 0;

			cr_phase_of(cfg, frac, &ph);

			if (have_prev && ph.first_rel == prev.first_rel && ph.last_rel == prev.last_rel && ph.table_at == prev.table_at)
				continue;

			prev = ph;
			have_prev = 1;

			if (ph.taps == 0 || ph.taps > 0x100000u)
				return fail(out, 1, "configuration yields an empty tap window (the reference would divide by zero)");

			if (ph.table_at + cfg->step * (uint64_t)(ph.taps - 1) >= table_len)
				return fail(out, 1, "configuration indexes outside the Lanczos table (the reference asserts, clownresampler.h:1012)");

			for (c = 0; c < ncombos; ++c)
				if (combos[c].mr == ph.first_rel && combos[c].xr == ph.last_rel)
					break;

			if (c == ncombos)
			{
				if (ncombos == 4)
					return fail(out, 1, "more than four (min_relative, max_relative) combinations");
				combos[c].mr = ph.first_rel;
				combos[c].xr = ph.last_rel;
				combos[c].klo = combos[c].khi = ph.table_at;
				++ncombos;
			}
			else
			{
				if (ph.table_at < combos[c].klo)
					combos[c].klo = ph.table_at;
				if (ph.table_at > combos[c].khi)
					combos[c].khi = ph.table_at;
			}

			if (ph.first_rel < mr_lo)
				mr_lo = ph.first_rel;
			if (ph.first_rel > mr_hi)
				mr_hi = ph.first_rel;
			if (ph.first_rel < off_lo)
				off_lo = ph.first_rel;
			if (ph.first_rel + ph.taps > off_hi)
				off_hi = ph.first_rel + ph.taps;

			for (t = 0; t < ph.taps; ++t)
			{
				const int64_t w = table[ph.table_at + cfg->step * t];

				sum += w;

				if (w > max_weight)
					max_weight = w;
				if (w < min_weight)
					min_weight = w;

				if (w != 0)
				{
					if (ph.first_rel + t < nz_lo)
						nz_lo = ph.first_rel + t;
					if (ph.first_rel + t + 1 > nz_hi)
						nz_hi = ph.first_rel + t + 1;
					if (t < rel_lo)
						rel_lo = t;
					if (t + 1 > rel_hi)
						rel_hi = t + 1;
				}
			}

			if (sum == 0)
				return fail(out, 1, "a phase has weight sum 0 (the reference divides by zero, clownresampler.h:1025)");
		}
	}

	/* from here on a failure only means "use the generic kernel".  The device's 32-bit row-index arithmetic needs
	   small scalars; the reference accepts stretches up to 4096 (clownresampler.h:974), far beyond any LDS window */
	if (cfg->skr >= (1u << 28) || cfg->step > 1024u || cfg->radius_frames > 4096u)
		return fail(out, 0, "configuration outside the fast kernel's 32-bit index range");

	if (nz_lo >= nz_hi) /* cannot happen once every sum is non-zero, but keep the window sane */
	{
		nz_lo = off_lo;
		nz_hi = off_hi;
	}

	out->first_slot = nz_lo;
	out->slots = nz_hi - nz_lo;
	out->row_stride = (out->slots + 1u + 3u) & ~3u;
	out->delta = (uint32_t)cfg->delta;
	out->skr = (uint32_t)cfg->skr;
	out->step = (uint32_t)cfg->step;

	/* ---- row-index form ---- */
	if (cfg->delta == 0 && cfg->step == 1024u)
	{
		out->row_mode = CRHIP_ROWMODE_UPSAMPLE;
		out->rows = (FRAC_ONE >> 6) + 1u;
	}
	else
	{
		/* combinations were met in order of increasing frac; each gets a contiguous block of rows */
		unsigned c;
		uint32_t next = 0;
		int64_t off[4];
		int64_t a = 0, b = 0, cc;

		if (ncombos > 3)
			return fail(out, 0, "four (min_relative, max_relative) combinations: not affine");

		for (c = 0; c < ncombos; ++c)
		{
			combos[c].base = next;
			next += (uint32_t)(combos[c].khi - combos[c].klo + 1);
			off[c] = (int64_t)combos[c].base - (int64_t)combos[c].klo;
		}

		for (c = 1; c < ncombos; ++c)
		{
			const int dm = (int)combos[c].mr - (int)combos[c - 1].mr;
			const int dx = (int)combos[c].xr - (int)combos[c - 1].xr;
			const int64_t d = off[c] - off[c - 1];

			if (dm == 1 && dx == 0)
				a = d;
			else if (dm == 0 && dx == 1)
				b = d;
			else if (dm == 1 && dx == 1)
				a = d; /* both step together: either coefficient can carry the difference */
			else
				return fail(out, 0, "unexpected (min_relative, max_relative) sequence");
		}

		cc = off[0] - a * combos[0].mr - b * combos[0].xr;

		if (a < -0x7FFFFFFF || a > 0x7FFFFFFF || b < -0x7FFFFFFF || b > 0x7FFFFFFF || cc < -0x7FFFFFFF || cc > 0x7FFFFFFF)
			return fail(out, 0, "row-index coefficients out of range");

		out->row_mode = CRHIP_ROWMODE_AFFINE;
		out->aff_a = (int32_t)a;
		out->aff_b = (int32_t)b;
		out->aff_c = (int32_t)cc;
		out->rows = next;

		/* SHIFTED windows.  The phases of a stretched kernel start at different frames (min_relative takes two values), so a
		   window common to all of them is one slot longer than the longest phase: 48 -> 44.1 kHz has 5-6 taps on 7 common
		   slots.  The affine row index already computes min_relative on the device, so every row is laid out from ITS OWN
		   first tap instead and the device starts a frame's window (min_relative - first_mr) frames later: one slot - a
		   seventh of cfg 4's multiplies - less.  The window a tile must hold grows by window_extra frames at its end. */
		if (rel_lo < rel_hi)
		{
			out->first_mr = mr_lo;
			out->window_extra = mr_hi - mr_lo;
			out->rel_first = rel_lo;
			out->first_slot = mr_lo + rel_lo;
			out->slots = rel_hi - rel_lo;
			out->row_stride = (out->slots + 1u + 3u) & ~3u;
			out->shifted = 1;
		}
		else
			return fail(out, 0, "no non-zero weight");   /* (cannot happen once every phase's sum is non-zero) */
	}

	if ((uint64_t)out->rows * out->row_stride > (1u << 24))
		return fail(out, 0, "polyphase table too large");

	/* ---- pass 2: fill and cross-check the rows ---- */
	{
		const size_t row_ints = out->row_stride;
		unsigned char *filled = (unsigned char *)calloc(out->rows, 1);
		int32_t *scratch = (int32_t *)malloc(row_ints * sizeof(int32_t));
		int ok = 1;
		int eligible = 1;
		const char *why = "";

		out->weights = (int32_t *)calloc((size_t)out->rows * row_ints, sizeof(int32_t));

		if (filled == NULL || scratch == NULL || out->weights == NULL)
		{
			free(filled);
			free(scratch);
			cr_poly_free(out);
			return fail(out, 1, "out of host memory");
		}

		/* The 32-bit kernels form sample * weight with the 24-bit multiplier (low 32 bits of the product) and truncate THAT:
		   the product must fit int32 for every sample in [-32768, 32767], i.e. -65536 < weight <= 65536 (the one product of
		   magnitude 2^31 that still fits is -32768 * 65536 = INT32_MIN).  The stock Lanczos tables peak at exactly 65536; a
		   caller-supplied table with larger weights takes the 64-bit generic kernel. */
		if (max_weight > 65536 || min_weight <= -65536)
		{
			eligible = 0;
			why = "a table weight outside (-65536, 65536]: sample * weight would not fit 32 bits";
		}

		{
		cr_phase prev;
		uint32_t prev_row = 0;
		int have_prev = 0;

		for (frac = 0; frac < FRAC_ONE && ok; ++frac)
		{
			cr_phase ph;
			uint32_t t, row;
			int64_t sum = 0, abs_sum = 0, recip, acc_bound;

			cr_phase_of(cfg, frac, &ph);
			row = cr_poly_row_of(out, frac);

			if (row >= out->rows)
			{
				ok = 0;
				break;
			}

			/* same taps as the previous fraction: it must land on the same row, whose contents were checked then */
			if (have_prev && ph.first_rel == prev.first_rel && ph.last_rel == prev.last_rel && ph.table_at == prev.table_at)
			{
				if (row != prev_row)
					ok = 0;
				continue;
			}

			prev = ph;
			prev_row = row;
			have_prev = 1;

			memset(scratch, 0, row_ints * sizeof(int32_t));

			for (t = 0; t < ph.taps; ++t)
			{
				const int64_t w = table[ph.table_at + cfg->step * t];
				const uint32_t offset = ph.first_rel + t;

				sum += w;
				abs_sum += w < 0 ? -w : w;

				if (w != 0)
				{
					/* slot of this tap: counted from the common window's first frame, or - shifted rows - from the phase's own */
					const uint32_t slot = out->shifted ? t - out->rel_first : offset - out->first_slot;

					if ((out->shifted ? t < out->rel_first : offset < out->first_slot) || slot >= out->slots)
					{
						ok = 0;
						break;
					}
					scratch[slot] = (int32_t)w;
				}
			}

			if (!ok)
				break;

			/* (cc_s32f)0x80000000 / sum with the reference's LP64 types: signed 64-bit, truncating (:1025) */
			recip = (int64_t)2147483648ll / sum;
			acc_bound = abs_sum / 2 + out->slots;

			if (recip <= 0 || recip >= (1 << 23))
			{
				eligible = 0;
				why = "a reciprocal is not a positive 23-bit number";
				recip = 0;
			}
			else if (acc_bound >= (1 << 23) || acc_bound * recip >= 4294967296ll)
			{
				eligible = 0;
				why = "accumulator * reciprocal may exceed 32 bits";
			}
			else if (acc_bound * recip >= 2147483648ll)
			{
				out->norm_mode = CRHIP_NORM_U32; /* e.g. the 8-lobe table: sum|w| can exceed 2 * sum(w) in a row */
			}

			scratch[out->slots] = (int32_t)recip;

			if (!filled[row])
			{
				memcpy(out->weights + (size_t)row * row_ints, scratch, row_ints * sizeof(int32_t));
				filled[row] = 1;
			}
			else if (memcmp(out->weights + (size_t)row * row_ints, scratch, row_ints * sizeof(int32_t)) != 0)
			{
				ok = 0; /* two fractions with different taps share a row: the index form does not fit */
			}
		}
		}

		free(filled);
		free(scratch);

		if (!ok)
		{
			cr_poly_free(out);
			return fail(out, 0, "row index form does not separate the phases");
		}

		out->eligible = eligible;
		out->reason = why;
	}

	return 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* LDS image of the rows                                                                                   */
/* ------------------------------------------------------------------------------------------------------- */

uint32_t cr_poly_plane_rows(const cr_poly *poly)
{
	return (poly->rows + 15u) & ~15u;
}

uint32_t cr_poly_phys_row(uint32_t row, uint32_t swizzle)
{
	return (row & ~15u) | ((row + swizzle * (row >> 4)) & 15u);
}

/* ds_read_b128 services a wave64 in four groups of 16 lanes; within a group, lanes whose 16-byte slots share a
   bank slot (address / 16 mod 16) serialise, identical addresses broadcast (MI355X_MICROARCH.md, LDS). */
static const unsigned char B128_GROUPS[4][16] = {
	{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
	{4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
	{32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
	{36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63},
};

/* which frame of its 64 a lane takes (crhip_poly_launch.lane_map) */
static unsigned frame_of_lane(unsigned lane, uint32_t lane_map)
{
	return lane_map ? (((lane & 31u) << 1) | (lane >> 5)) : lane;
}

/* Modelled extra LDS cycles per row read (ds_read_b128) of a wave, for all 16 rotations at once: cost[k] for swizzle k.  The rows
   of the 64 lanes are worked out ONCE per trial and every rotation is scored on them - a plan that differs from its siblings only
   in its increment is made per distinct ratio of a variable-rate stream, and scoring the rotations one by one (16 x 32 x 64 row
   indices) was most of the 120 us such a plan cost (tools/plan_create_rate.py). */
static void swizzle_costs(const cr_poly *poly, uint64_t increment, uint32_t lane_map, double cost[16])
{
	/* lanes of a wave hold the output frames of one block of 64: fraction of lane l = frac0 + frame(l) * increment (mod 65536) */
	unsigned extra[16] = {0};
	unsigned trial, k;

	for (trial = 0; trial < 32; ++trial)
	{
		const uint32_t frac0 = (trial * 40503u + 977u) & 0xFFFFu;
		uint32_t rows[64];
		unsigned g, lane;

		for (lane = 0; lane < 64; ++lane)
			rows[lane] = cr_poly_row_of(poly, (uint32_t)((frac0 + (uint64_t)frame_of_lane(lane, lane_map) * increment) & 0xFFFFu));

		for (g = 0; g < 4; ++g)
		{
			/* the DISTINCT rows of the group (identical addresses broadcast; a rotation is a bijection, so distinct rows stay
			   distinct under every one of them): once per group, not once per rotation */
			uint32_t distinct[16];
			unsigned n = 0, i, j;

			for (i = 0; i < 16; ++i)
			{
				const uint32_t row = rows[B128_GROUPS[g][i]];
				int seen = 0;

				for (j = 0; j < n; ++j)
					if (distinct[j] == row)
						seen = 1;
				if (!seen)
					distinct[n++] = row;
			}

			for (k = 0; k < 16; ++k)
			{
				unsigned count[16] = {0};
				unsigned worst = 0;

				for (i = 0; i < n; ++i)
				{
					const unsigned slot = cr_poly_phys_row(distinct[i], k) & 15u;

					if (++count[slot] > worst)
						worst = count[slot];
				}
				extra[k] += worst - 1;
			}
		}
	}

	for (k = 0; k < 16; ++k)
		cost[k] = extra[k] / 32.0;
}

void cr_poly_slot_signs(const cr_poly *poly, uint32_t *positive, uint32_t *negative)
{
	uint32_t row, slot;

	*positive = 0;
	*negative = 0;

	for (row = 0; row < poly->rows; ++row)
	{
		for (slot = 0; slot < poly->slots && slot < 32u; ++slot)
		{
			const int32_t w = poly->weights[(size_t)row * poly->row_stride + slot];

			if (w > 0)
				*positive |= 1u << slot;
			else if (w < 0)
				*negative |= 1u << slot;
		}
	}
}

uint32_t cr_poly_slots_reaching(const cr_poly *poly, int32_t magnitude)
{
	uint32_t row, slot, mask = 0;

	for (row = 0; row < poly->rows; ++row)
	{
		for (slot = 0; slot < poly->slots && slot < 32u; ++slot)
		{
			const int32_t w = poly->weights[(size_t)row * poly->row_stride + slot];

			if (w >= magnitude || w <= -magnitude)
				mask |= 1u << slot;
		}
	}

	return mask;
}

/* Modelled extra LDS cycles per read of one window slot by a wave of k_wave2 (the expanded window: one dword per sample, a lane's
   frame at CH dwords per input frame).  An even channel count reads pairs of channels as ds_read_b64 - two groups of 32 lanes,
   64 banks, a lane on two of them -, an odd one single dwords as ds_read_b32 - 32 banks (MI355X_MICROARCH.md, LDS).  Within a
   group every further distinct address on a bank costs a cycle. */
double cr_window_conflicts(const cr_poly *poly, const cr_config *cfg, uint64_t increment, uint32_t channels, uint32_t lane_map)
{
	const unsigned banks = (channels % 2u == 0u) ? 64u : 32u, width = (channels % 2u == 0u) ? 2u : 1u;
	double extra = 0.0;
	unsigned trial;

	for (trial = 0; trial < 32; ++trial)
	{
		const uint32_t frac0 = (trial * 40503u + 977u) & 0xFFFFu;
		unsigned g;

		for (g = 0; g < 2; ++g)
		{
			uint64_t address[32];
			unsigned count[64] = {0};
			unsigned i, k, w, worst = 0;

			for (i = 0; i < 32; ++i)
			{
				const uint64_t rel = frac0 + (uint64_t)frame_of_lane(32u * g + i, lane_map) * increment;
				cr_phase phase;

				cr_phase_of(cfg, (uint32_t)(rel & 0xFFFFu), &phase);
				address[i] = ((rel >> 16) + (poly->shifted ? phase.first_rel - poly->first_mr : 0u)) * channels;
			}
			for (i = 0; i < 32; ++i)
			{
				int seen = 0;

				for (k = 0; k < i; ++k)
					if (address[k] == address[i])
						seen = 1;   /* same address: broadcast */
				if (seen)
					continue;
				for (w = 0; w < width; ++w)
					if (++count[(address[i] + w) % banks] > worst)
						worst = count[(address[i] + w) % banks];
			}
			extra += worst > 0 ? worst - 1 : 0;
		}
	}
	return extra / 32.0;
}

uint32_t cr_poly_pick_swizzle(const cr_poly *poly, uint64_t increment, double *conflict_plain, double *conflict_best)
{
	return cr_poly_pick_swizzle_mapped(poly, increment, 0u, conflict_plain, conflict_best);
}

uint32_t cr_poly_pick_swizzle_mapped(const cr_poly *poly, uint64_t increment, uint32_t lane_map, double *conflict_plain, double *conflict_best)
{
	uint32_t best = 0, k;
	double cost[16], best_cost;

	swizzle_costs(poly, increment, lane_map, cost);
	best_cost = cost[0];

	if (conflict_plain != NULL)
		*conflict_plain = best_cost;

	for (k = 1; k < 16; ++k)
	{
		if (cost[k] < best_cost - 1e-9)
		{
			best_cost = cost[k];
			best = k;
		}
	}

	if (conflict_best != NULL)
		*conflict_best = best_cost;

	return best;
}

int32_t *cr_poly_device_image(const cr_poly *poly, uint32_t swizzle, int layout, uint32_t *device_row_stride)
{
	const uint32_t plane_rows = cr_poly_plane_rows(poly);
	const uint32_t weight_planes = (poly->slots + 3u) / 4u;
	const uint32_t stride = layout == CR_IMAGE_SPLIT ? 4u * (weight_planes + 1u) : poly->row_stride;
	const uint32_t planes = stride / 4u;
	int32_t *image = (int32_t *)calloc((size_t)plane_rows * stride, sizeof(int32_t));
	uint32_t row, q, e;

	*device_row_stride = stride;

	if (image == NULL)
		return NULL;

	for (row = 0; row < poly->rows; ++row)
	{
		const uint32_t phys = cr_poly_phys_row(row, swizzle);
		const int32_t *src = poly->weights + (size_t)row * poly->row_stride;

		for (q = 0; q < planes; ++q)
		{
			for (e = 0; e < 4; ++e)
			{
				const uint32_t index = 4u * q + e;
				int32_t value;

				if (layout == CR_IMAGE_SPLIT)
					value = q < weight_planes ? (index < poly->slots ? src[index] : 0) : (e == 0 ? src[poly->slots] : 0);
				else
					value = src[index];

				image[((size_t)q * plane_rows + phys) * 4u + e] = value;
			}
		}
	}

	return image;
}

/* ------------------------------------------------------------------------------------------------------- */

uint64_t cr_count_output_frames(uint64_t pos_int, uint64_t pos_frac, uint64_t increment, uint64_t total_input_frames)
{
	/* frames are emitted while position_integer < total (clownresampler.h:1063), i.e. while
	   P0 + j * increment < total * 65536 with P0 = pos_int * 65536 + pos_frac */
	const unsigned __int128 start = (unsigned __int128)pos_int * FRAC_ONE + pos_frac;
	const unsigned __int128 limit = (unsigned __int128)total_input_frames * FRAC_ONE;

	if (start >= limit || increment == 0)
		return 0;

	return (uint64_t)((limit - start + increment - 1) / increment);
}

void cr_advance(uint64_t *pos_int, uint64_t *pos_frac, uint64_t increment, uint64_t frames)
{
	/* clownresampler.h:1076-1078 applied `frames` times */
	const unsigned __int128 fr = (unsigned __int128)*pos_frac + (unsigned __int128)increment * frames;

	*pos_int += (uint64_t)(fr / FRAC_ONE);
	*pos_frac = (uint64_t)(fr % FRAC_ONE);
}

uint64_t cr_input_extent(const cr_config *cfg, uint64_t pos_int, uint64_t pos_frac, uint64_t increment, uint64_t frames)
{
	/* the last frame's window ends at position_integer + radius_frames + max_relative (clownresampler.h:996),
	   max_relative <= radius_frames (:1004) */
	uint64_t pi = pos_int, pf = pos_frac;

	if (frames == 0)
		return 0;

	cr_advance(&pi, &pf, increment, frames - 1);
	return pi + 2 * cfg->radius_frames;
}

uint64_t cr_hash_bytes(const void *data, size_t bytes, uint64_t seed)
{
	/* FNV-1a over 8-byte words (tail bytewise); only used as a cache key.  Every host-buffer call hashes the caller's 48 KB
	   table, so the bulk runs as four independent chains (the multiply's latency, not its throughput, is what a single
	   chain pays): ~2 us instead of ~8 us per call. */
	const unsigned char *p = (const unsigned char *)data;
	uint64_t h = seed ^ 1469598103934665603ull;
	size_t i = 0;

	if (bytes >= 64)
	{
		uint64_t a = h, b = h ^ 0x9E3779B97F4A7C15ull, c = h ^ 0xC2B2AE3D27D4EB4Full, d = h ^ 0x165667B19E3779F9ull;

		for (; i + 32 <= bytes; i += 32)
		{
			uint64_t w[4];
			memcpy(w, p + i, 32);
			a = (a ^ w[0]) * 1099511628211ull;
			b = (b ^ w[1]) * 1099511628211ull;
			c = (c ^ w[2]) * 1099511628211ull;
			d = (d ^ w[3]) * 1099511628211ull;
		}
		h = (a ^ (b >> 7)) * 1099511628211ull;
		h = (h ^ (c >> 11)) * 1099511628211ull;
		h = (h ^ (d >> 13)) * 1099511628211ull;
	}

	for (; i + 8 <= bytes; i += 8)
	{
		uint64_t w;
		memcpy(&w, p + i, 8);
		h = (h ^ w) * 1099511628211ull;
	}

	for (; i < bytes; ++i)
		h = (h ^ p[i]) * 1099511628211ull;

	return h;
}
