// pnode_amd -- internal helpers shared by the kernel launchers and the host engine.
#pragma once
#include <string>

namespace pn {
// records the message for pn_last_error() and returns 1
int fail(const std::string &msg);
}  // namespace pn
