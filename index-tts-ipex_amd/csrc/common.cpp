#include "itts_common.h"
#include <mutex>
namespace itts {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* last_error() { return g_err.c_str(); }
}  // namespace itts
