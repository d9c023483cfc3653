// Force-included when compiling the reference's DCSEncoder.cpp with g++ (fixture generator only, build container
// only): the handful of MSVC CRT names that file uses.  Nothing of the decode path is touched by this.
#pragma once
#include <cerrno>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#ifndef _countof
#define _countof(a) (sizeof(a) / sizeof((a)[0]))
#endif
static inline int fopen_s(FILE **fp, const char *name, const char *mode) { *fp = fopen(name, mode); return *fp ? 0 : errno; }
static inline int _vscprintf(const char *fmt, va_list va) { va_list c; va_copy(c, va); const int n = vsnprintf(nullptr, 0, fmt, c); va_end(c); return n; }
#define vsprintf_s(buf, size, fmt, va) vsnprintf(buf, size, fmt, va)
