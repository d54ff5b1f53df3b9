// asan_host_shim.cpp -- TEST INFRASTRUCTURE: the two symbols the product's host parsers take from lutr_api.cpp
// (which needs HIP), so that cube_parse.cpp + lut_formats.cpp can be built alone under gcc's sanitizers
// (oracle/Makefile `asan`).  Never linked into liblutr.so.
#include <cstdarg>
#include <cstdio>
#include <string>

#include "lutr_internal.h"

static thread_local std::string g_err;

namespace lutr {
void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace lutr

extern "C" const char *lutr_last_error(void) { return g_err.c_str(); }
