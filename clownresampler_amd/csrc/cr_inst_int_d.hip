// cr_inst_int_d.hip - an instance unit of k_int (cr_kint.hpp): the output-stationary instances.  PERIODIC ratios - the increment repeats after 2 or 4 output frames
// (3:2; also 1:2 and 1:4, see below), so a launch uses 2 or 4 rows in turn and they travel in the kernel arguments like the one row of a whole-number
// ratio.  The constants are what tools/int_shapes.py prints for the reference's 3-lobe table at fractional position 0; the host
// checks every launch's rows, window starts and zero slots against them (cr_context.c int_launch_row) - anything else takes the
// plan's ordinary kernel.
#include "cr_inst_int.hpp"

const void *crhip_int_instances_d(int *count)
{
	static const int_instance table[] = {
	    // 3:2 (48 -> 32 kHz, 96 -> 64 kHz): 9 slots, phase 1 starts one frame later; the last slot of phase 0 is zero.  16 frames per
	    // lane both, by measurement (stereo 8 / 16 / 24 frames per lane: 0.53 / 0.67 / 0.62 of the roofline, mono 16 / 32 / 48: 0.54 / 0.45 /
	    // 0.53; profiles/r03_kint_periodic.log.  Not the bank conflicts of the staged stores - a conflict-free stage stride, wrong
	    // results, timed the same)
	    make_per<2, 3, 2, 0x100u, 9, 16, 0x18846ull, 0x0ull, 0x100ull>(),   // stereo: lane stride 96 B
	    make_per<1, 3, 2, 0x100u, 9, 16, 0x18846ull, 0x0ull, 0x100ull>(),   // mono: 48 B
	    // 2:1 with the 8-lobe and the 5-lobe table (CLOWNRESAMPLER_KERNEL_RADIUS 8 / 5: 32 / 20 slots), output-stationary order
	    make_int_long<2, 8, 2, 6>(),    // stereo, 8 lobes: lane stride 48 B, 42 input frames per lane
	    make_int_long<1, 8, 2, 12>(),   // mono, 8 lobes: 48 B, 54 input frames
	    make_int_long<2, 5, 2, 6>(),    // stereo, 5 lobes
	    make_int_long<1, 5, 2, 12>(),   // mono, 5 lobes
	    // 3:1 (48 / 30 slots) and, with 5 lobes, 4:1 (40 slots)
	    make_int_long<2, 8, 3, 4>(),    // stereo, 8 lobes 3:1: 48 B, 57 input frames per lane
	    make_int_long<1, 8, 3, 8>(),    // mono
	    make_int_long<2, 5, 3, 4>(),    // stereo, 5 lobes 3:1
	    make_int_long<1, 5, 3, 8>(),    // mono
	    make_int_long<2, 5, 4, 3>(),    // stereo, 5 lobes 4:1
	    make_int_long<1, 5, 4, 6>(),    // mono
	    // periodic ratios with the 8- and 5-lobe tables (tools/int_shapes.py 8 / 5 ...): 1:2 - phase 0 is the input sample, phase 1 a full
	    // row of 15 / 9 slots: eight / five taps per frame on average, which DO pay for the staging - and 3:2 (24 / 15 slots)
	    make_per<2, 1, 2, 0x0u, 15, 24, 0x152A8000ull, 0x80ull, 0x7F7Full>(),                // 8 lobes 1:2 stereo (lane stride 48 B; 8 frames per lane, 16 B: 0.51 against 0.66)
	    make_per<1, 1, 2, 0x0u, 15, 48, 0x152A8000ull, 0x80ull, 0x7F7Full>(),                // ... mono (48 B; 16 frames per lane: 0.51 against 0.57)
	    make_per<2, 3, 2, 0x200u, 24, 8, 0x1246DBB6C492ull, 0x0ull, 0x800000000001ull>(),    // 8 lobes 3:2 stereo
	    make_per<1, 3, 2, 0x200u, 24, 16, 0x1246DBB6C492ull, 0x0ull, 0x800000000001ull>(),   // ... mono
	    make_per<2, 1, 2, 0x0u, 9, 24, 0x29400ull, 0x10ull, 0x1EFull>(),                     // 5 lobes 1:2 stereo (0.53 -> 0.67)
	    make_per<1, 1, 2, 0x0u, 9, 48, 0x29400ull, 0x10ull, 0x1EFull>(),                     // ... mono (0.51 -> 0.60)
	    make_per<2, 3, 2, 0x100u, 15, 16, 0x1B121236ull, 0x0ull, 0x4000ull>(),               // 5 lobes 3:2 stereo
	    make_per<1, 3, 2, 0x100u, 15, 16, 0x1B121236ull, 0x0ull, 0x4000ull>(),               // ... mono
	    make_per<2, 1, 4, 0x0u, 15, 16, 0x54AAA95552A8000ull, 0x80ull, 0x7F7Full>(),         // 8 lobes 1:4 stereo (60 weights)
	    make_per<1, 1, 4, 0x0u, 15, 16, 0x54AAA95552A8000ull, 0x80ull, 0x7F7Full>(),         // ... mono (lane stride 8 B)
	    make_per<2, 1, 4, 0x0u, 9, 16, 0xA552A9400ull, 0x10ull, 0x1EFull>(),                 // 5 lobes 1:4 stereo
	    make_per<1, 1, 4, 0x0u, 9, 16, 0xA552A9400ull, 0x10ull, 0x1EFull>(),                 // ... mono
	    // 3 lobes 1:2, MONO only (8 -> 16, 22.05 -> 44.1, 24 -> 48 kHz): 48 frames per lane, 0.60 against k_poly's 0.57
	    make_per<1, 1, 2, 0x0u, 5, 48, 0x240ull, 0x4ull, 0x1Bull>(),
	    // NOT instantiated: stereo upsampling by 2, and by 4, with the 3-lobe table (tools/int_shapes.py 3 24000:48000 12000:48000 prints their constants:
	    // make_per<CH, 1, 2, 0x0u, 5, K, 0x240ull, 0x4ull, 0x1Bull>, make_per<CH, 1, 4, 0x0u, 5, K, 0x94A40ull, 0x4ull, 0x1Bull> -
	    // phase 0 is the input sample itself).  Bit-exact, and slower than k_poly there (1:2 stereo 0.53-0.65 against 0.70-0.71 with
	    // 8 / 16 / 24 frames per lane; 1:4 0.49-0.57 / 0.28-0.31 against 0.62-0.63 / 0.45-0.49; profiles/r03_kint_periodic.log): three
	    // taps per frame on average do not
	    // pay for staging every frame through LDS.
	};
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}
