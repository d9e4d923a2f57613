// cr_inst_runtime_w.hip - instance unit: k_wave2 with a run-time slot count (any downsampling configuration, 1 to 8 channels, both
// output forms): what plans WITHOUT a specialised instance run when their windows are long enough for the 2-instruction tap
// to pay and fit a wave's slice of the LDS (cr_context.c plan_geometry)  (see cr_instances.hpp)
#include "cr_instances.hpp"
// k_wave2s (a lane per channel PAIR: cr_kwave2s.hpp) never became a default - it measured slower than the run-time-slot k_poly for the wide
// frames it was written for (profiles/r03_wave2s.log) - so it is NOT in the default build any more (a kernel nothing selects,
// compiled into both libraries): `make CRA_CFLAGS=-DCRA_WITH_WAVE2S` brings it back behind variant 32 for comparison
// (tests/test_gpu_parity.py::test_channel_pair_per_lane_kernel then runs instead of skipping).
#ifdef CRA_WITH_WAVE2S
#include "cr_kwave2s.hpp"
#endif

namespace
{
template <int OUT16>
poly_fn pick(uint32_t channels)
{
	switch (channels)
	{
#define CRK_CASE(CH) case CH: return (poly_fn)k_wave2<CH, 0, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 16, 1, 1, OUT16, 1, 0u, 1>;
		CRK_CASE(1) CRK_CASE(2) CRK_CASE(3) CRK_CASE(4) CRK_CASE(5) CRK_CASE(6) CRK_CASE(7) CRK_CASE(8)
#undef CRK_CASE
		default: return nullptr;
	}
}
} // namespace

namespace crk
{
void *runtime_wave2s_instance(uint32_t channels, int out16)
{
#ifdef CRA_WITH_WAVE2S
	if (channels < 3 || channels > CRHIP_MAX_CHANNELS)
		return nullptr;
	if (channels % 2 == 0)
		return out16 ? (void *)k_wave2s<1, 1, 1> : (void *)k_wave2s<1, 0, 1>;
	return out16 ? (void *)k_wave2s<0, 1, 1> : (void *)k_wave2s<0, 0, 1>;
#else
	(void)channels;
	(void)out16;
	return nullptr;
#endif
}

void *runtime_wave2_instance(uint32_t channels, uint32_t mode, int out16)
{
	if (mode != CRHIP_ROWMODE_AFFINE)
		return nullptr;
	return out16 ? (void *)pick<1>(channels) : (void *)pick<0>(channels);
}
} // namespace crk
