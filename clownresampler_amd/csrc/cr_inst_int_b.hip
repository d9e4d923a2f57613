// cr_inst_int_b.hip - an instance unit of k_int (cr_kint.hpp): three to five channels.  (Its own unit for the build's sake: an instance is a
// fully unrolled tile body, K x slots x channels taps.)  The channel pairs of a frame run one after the other over the same window
// registers; an odd count's last channel is alone on its pass.
#include "cr_inst_int.hpp"

const void *crhip_int_instances_b(int *count)
{
	static const int_instance table[] = {
	    make_int<4, 3, 2, 5>(),   // 4 channels 2:1: 80 B
	    make_int<4, 3, 3, 6>(),   // 4 channels 3:1: 144 B
	    make_int<4, 3, 4, 4>(),   // 4 channels 4:1: 128 B (even multiple)
	    make_int<4, 3, 6, 2>(),   // 4 channels 6:1: 96 B
	    make_int<3, 3, 2, 12>(),  // 3 channels 2:1: 144 B
	    make_int<3, 3, 3, 8>(),   // 3 channels 3:1: 144 B
	    make_int<3, 3, 4, 6>(),   // 3 channels 4:1: 144 B
	    make_int<5, 3, 2, 4>(),   // 5 channels 2:1: 80 B
	    make_int<5, 3, 4, 2>(),   // 5 channels 4:1: 80 B
	    make_int<5, 3, 3, 4>(),   // 5 channels 3:1: 120 B
	    make_int<3, 3, 6, 2>(),   // 3 channels 6:1: 72 B
	    make_int<5, 3, 6, 2>(),   // 5 channels 6:1: 120 B
	};
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}
