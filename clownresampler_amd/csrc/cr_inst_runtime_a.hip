// cr_inst_runtime_a.hip - instance unit: run-time slot count, 1 to 4 channels  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace
{
template <int OUT16>
poly_fn pick(uint32_t channels, uint32_t mode, uint32_t norm)
{
	switch (channels)
	{
		case 1: return pick_runtime<1, OUT16>(mode, norm);
		case 2: return pick_runtime<2, OUT16>(mode, norm);
		case 3: return pick_runtime<3, OUT16>(mode, norm);
		case 4: return pick_runtime<4, OUT16>(mode, norm);
		default: return nullptr;
	}
}
} // namespace

namespace crk
{
void *runtime_instance_1_4(uint32_t channels, uint32_t mode, uint32_t norm, int out16)
{
	return out16 ? (void *)pick<1>(channels, mode, norm) : (void *)pick<0>(channels, mode, norm);
}
} // namespace crk
