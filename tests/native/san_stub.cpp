// The two symbols csrc/wavio.hip takes from the rest of libdcs (api.hip), for the host-only sanitizer builds of tests/test_sanitizers_cpu.py.
#include <stdarg.h>
#include <stdio.h>
void dcs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
