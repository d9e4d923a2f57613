/*
 * flac2pcm.c - decode a FLAC file to raw interleaved s16 little-endian PCM with
 * the reference's vendored dr_flac (included in place from /root/reference/tests,
 * -DDRFLAC_HEADER='"/root/reference/tests/dr_flac.h"'), exactly as the reference
 * harness does (tests/test-low-level.c:93,148: drflac_open_file +
 * drflac_read_pcm_frames_s16).  Container-only fixture generator; TEST
 * INFRASTRUCTURE, never shipped.  Prints "channels rate frames" on stdout.
 */
#include <stdio.h>
#include <stdlib.h>

#define DR_FLAC_IMPLEMENTATION
#define DR_FLAC_NO_OGG
#define DRFLAC_API static
#include DRFLAC_HEADER

int main(int argc, char **argv)
{
	drflac *dec;
	drflac_int16 *pcm;
	drflac_uint64 got;
	FILE *out;

	if (argc < 3)
	{
		fprintf(stderr, "usage: %s in.flac out.s16\n", argv[0]);
		return 1;
	}

	dec = drflac_open_file(argv[1], NULL);
	if (dec == NULL)
		return 2;

	pcm = (drflac_int16 *)malloc((size_t)dec->totalPCMFrameCount * dec->channels * sizeof(*pcm));
	got = drflac_read_pcm_frames_s16(dec, dec->totalPCMFrameCount, pcm);

	out = fopen(argv[2], "wb");
	if (out == NULL)
		return 3;
	fwrite(pcm, sizeof(*pcm), (size_t)got * dec->channels, out);
	fclose(out);

	printf("%u %u %llu\n", (unsigned)dec->channels, (unsigned)dec->sampleRate, (unsigned long long)got);
	drflac_close(dec);
	free(pcm);
	return 0;
}
