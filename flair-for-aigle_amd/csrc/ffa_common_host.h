// Host-only pieces shared by the non-kernel translation units of libflairhip.
#pragma once
#include "../../include/flairhip.h"

void ffa_set_error(const char* fmt, ...);
