// sq_error.h — thread-local last-error string behind sq_last_error() (include/squigly_hip.h).
#pragma once
#include <cstdarg>
#include <cstdio>

inline char* sq_error_buffer() { static thread_local char buf[1024] = { 0 }; return buf; }
// Formats the message, stores it, and returns 1 so callers can `return sq_set_error(...)`.
inline int sq_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    std::vsnprintf(sq_error_buffer(), 1024, fmt, ap);
    va_end(ap);
    return 1;
}
