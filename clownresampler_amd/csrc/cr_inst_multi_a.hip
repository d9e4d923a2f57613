// cr_inst_multi_a.hip - instance unit: 8 channels 48 -> 44.1 kHz with every tuning variant (BASELINE configs[3])  (see cr_instances.hpp)
#include "cr_instances.hpp"

namespace crk
{

int specials_multi_a(void *table, int capacity)
{
	static const special mine[] = {
	    // cfg 4: 8 channels 48 -> 44.1 kHz (5-6 taps; 6 slots on shifted windows).  Ticketed tiles since the tickets are scalar atomics and
	    // the mailbox an LDS word (round 1, with a vmcnt(0) behind either, measured them 1-8 % SLOWER here): 257.9 -> 243.7 us on one box
	    with_signed_chain<8, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 2, true>(make_special<8, 6, CRHIP_ROWMODE_AFFINE, CRHIP_NORM_S31, 2, false, true>()),
	};
	const int n = (int)(sizeof(mine) / sizeof(mine[0]));
	if (table == nullptr)
		return n;   // (asked for the count: specials() sizes its table from the providers)
	if (n > capacity)
		return -1;
	memcpy(table, mine, sizeof(mine));
	return n;
}

} // namespace crk
