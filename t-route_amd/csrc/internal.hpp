// internal.hpp -- shared between the translation units of libtrmc.so; not part of the C ABI.
#pragma once
#include <string>

namespace trmc {
// stores `msg` as the calling thread's trmc_last_error() and returns `code`
int fail_with(int code, const std::string &msg);
} // namespace trmc
