// libsfmi: version + small host utilities of the C ABI (include/sfmi.h).
#include "sfmi_common.h"

extern "C" {
int sfmi_version(void) { return 100; }
}
