// cr_inst_runtime_b.hip - instance unit: run-time slot count, 5 to 8 channels  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace
{
template <int OUT16>
poly_fn pick(uint32_t channels, uint32_t mode, uint32_t norm)
{
	switch (channels)
	{
		case 5: return pick_runtime<5, OUT16>(mode, norm);
		case 6: return pick_runtime<6, OUT16>(mode, norm);
		case 7: return pick_runtime<7, OUT16>(mode, norm);
		case 8: return pick_runtime<8, OUT16>(mode, norm);
		default: return nullptr;
	}
}
} // namespace

namespace crk
{
void *runtime_instance_5_8(uint32_t channels, uint32_t mode, uint32_t norm, int out16)
{
	return out16 ? (void *)pick<1>(channels, mode, norm) : (void *)pick<0>(channels, mode, norm);
}
} // namespace crk
