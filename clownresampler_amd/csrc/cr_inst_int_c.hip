// cr_inst_int_c.hip - an instance unit of k_int (cr_kint.hpp): six to sixteen channels.  (Its own unit for the build's sake: an instance is a
// fully unrolled tile body, K x slots x channels taps.)  The channel pairs of a frame run one after the other over the same window
// registers; an odd count's last channel is alone on its pass.
#include "cr_inst_int.hpp"

const void *crhip_int_instances_c(int *count)
{
	static const int_instance table[] = {
	    make_int<6, 3, 2, 6>(),   // 6 channels 2:1: 144 B
	    make_int<6, 3, 3, 4>(),   // 6 channels 3:1: 144 B
	    make_int<8, 3, 2, 4>(),   // 8 channels 2:1: 128 B (even multiple)
	    make_int<6, 3, 4, 1>(),   // 6 channels 4:1: 48 B
	    make_int<8, 3, 3, 1>(),   // 8 channels 3:1: 48 B
	    make_int<8, 3, 4, 1>(),   // 8 channels 4:1: 64 B
	    make_int<7, 3, 2, 4>(),   // 7 channels 2:1: 112 B
	    make_int<7, 3, 3, 2>(),   // 7 channels 3:1: 84 B
	    make_int<7, 3, 4, 2>(),   // 7 channels 4:1: 112 B
	    make_int<6, 3, 6, 1>(),   // 6 channels 6:1: 72 B
	    make_int<7, 3, 6, 2>(),   // 7 channels 6:1: 168 B
	    make_int<8, 3, 6, 1>(),   // 8 channels 6:1: 96 B
	    // nine to sixteen channels (the reference's maximum) at 2:1: one frame per lane
	    make_int<9, 3, 2, 4>(),   // 9 channels: 144 B
	    make_int<11, 3, 2, 2>(),  // 11 channels: 88 B (8-byte window reads)
	    make_int<13, 3, 2, 2>(),  // 13 channels: 104 B
	    make_int<15, 3, 2, 2>(),  // 15 channels: 120 B
	    make_int<10, 3, 2, 2>(),  // 10 channels: 80 B (four frames per lane, 160 B: 0.57 against 0.66)
	    make_int<12, 3, 2, 1>(),  // 12 channels: lane stride 48 B
	    make_int<14, 3, 2, 2>(),  // 14 channels: 112 B
	    make_int<16, 3, 2, 1>(),  // 16 channels: 64 B (two frames per lane, 128 B: 0.54 against 0.62)
	};
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}
