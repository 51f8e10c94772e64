// hrx_error.hpp — the thread-local text behind hrx_last_error(), for the translation units of the C ABI (hrx_api.cpp owns it).
#pragma once
#include <string>

namespace hrx {
// records `msg` as the calling thread's last error and returns `code`
int set_last_error(int code, const std::string &msg);
}  // namespace hrx
