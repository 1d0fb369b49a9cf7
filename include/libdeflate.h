/*
 * libdeflate.h - forwarder so that code which includes the reference header
 * by name builds against the MI355X engine.  The declarations live in
 * libdeflate_amd.h (same 21 entry points as /root/reference/libdeflate.h).
 */
#ifndef LIBDEFLATE_H
#define LIBDEFLATE_H
#include "libdeflate_amd.h"
#endif
