// cr_inst_runtime_c.hip - instance unit: run-time slot count, 9 to 16 channels (two lanes per frame)  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace
{
template <int OUT16>
poly_fn pick(uint32_t channels, uint32_t mode, uint32_t norm, uint32_t padded)
{
	switch (channels)
	{
		case 9: return pick_runtime_split<5, OUT16, 1>(mode, norm, padded);
		case 10: return pick_runtime_split<5, OUT16>(mode, norm, padded);
		case 11: return pick_runtime_split<6, OUT16, 1>(mode, norm, padded);
		case 13: return pick_runtime_split<7, OUT16, 1>(mode, norm, padded);
		case 15: return pick_runtime_split<8, OUT16, 1>(mode, norm, padded);
		case 12: return pick_runtime_split<6, OUT16>(mode, norm, padded);
		case 14: return pick_runtime_split<7, OUT16>(mode, norm, padded);
		case 16: return pick_runtime_split<8, OUT16>(mode, norm, padded);
		default: return nullptr;
	}
}
} // namespace

namespace crk
{
void *runtime_instance_9_16(uint32_t channels, uint32_t mode, uint32_t norm, int out16, uint32_t padded)
{
	return out16 ? (void *)pick<1>(channels, mode, norm, padded) : (void *)pick<0>(channels, mode, norm, padded);
}
} // namespace crk
