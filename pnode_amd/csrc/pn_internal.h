// pnode_amd -- internal helpers shared by the kernel launchers and the host engine.
#pragma once
#include <string>

namespace pn {
// records the message for pn_last_error() and returns 1
int fail(const std::string &msg);
// For launchers outside pn_kernels.hip.  0: profiling is off (launch plainly); 1: *e0 / *e1 are the start / stop events to hand
// to hipExtLaunchKernelGGL, the record is booked under kernel id `kid` with `bytes`; -1: failure (pn_last_error()).
int prof_events(int kid, double bytes, void **e0, void **e1);     // (hipEvent_t: this header is also read by plain g++)
}  // namespace pn
