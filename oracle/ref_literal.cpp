// TEST INFRASTRUCTURE ONLY — the reference's next_event.cpp as its own program, unmodified and with
// the C library's rand(): the literal `<width> <height> <iterations> <thread count>` run, in which
// every worker thread draws from glibc's one locked generator (next_event.cpp:19-22, 701-704).
// Timed beside the lock-free figure as the labelled "literal threads" CPU number; it is never a
// parity partner (its stream cannot be reproduced).  Built by oracle/Makefile (`make ref-literal`)
// from the source where it lies; outputs only under oracle/_ref/.
#include "shim/ptor_boost_shim.hpp"

#ifndef PTOR_REF_NEE
#error "define PTOR_REF_NEE as the quoted path of next_event.cpp in the reference checkout"
#endif
#include PTOR_REF_NEE
