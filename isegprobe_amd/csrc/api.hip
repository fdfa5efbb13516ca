#include "isp_common.h"
extern "C" int isp_abi_version(void) { return 19; }
