/* cr_buildid.c - the library's source id (csrc/Makefile: sha256 over its sources, first 16 hex digits). */
#include "source_id.h"

const char *ClownResamplerAMD_BuildId(void)
{
	return CRA_SOURCE_ID;
}

/* the kernel radii this build carries (csrc/Makefile RADII), space-separated */
const char *ClownResamplerAMD_BuiltRadii(void)
{
	return CRA_BUILT_RADII;
}
