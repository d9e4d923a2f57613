/* cr_buildid.c - the library's source id (csrc/Makefile: sha256 over its sources, first 16 hex digits). */
#include "source_id.h"

const char *ClownResamplerAMD_BuildId(void)
{
	return CRA_SOURCE_ID;
}
