// errors.hpp -- the reference's error transport (src/libs/errors/errors/error_handler.hpp:9-92)
// for the host code that drives the HIP library: exceptions by default, a returned struct under
// NO_EXCEPTIONS.  RND_CHECK turns a non-zero return of the C ABI into RAISE_ERROR_FATAL with the
// library's message, the way the samples treat a failed VkResult (dummy_obj_viewer_app.cpp:387-389).
#pragma once
#include <cstdint>
#include <cstdio>
#include <exception>
#include <string>

#include "../../include/renderer_amd.h"

namespace errors {
enum state { OK, WARN, FATAL };

#ifndef NO_EXCEPTIONS
class error : public std::exception {
public:
    explicit error(state s, int32_t c = 0, const std::string& m = "") : err_state(s), code((uint32_t)c), m_message(m) {}
    const char* what() const noexcept override { return m_message.c_str(); }
    state err_state{OK};
    uint32_t code{0};
private:
    std::string m_message;
};
#else
struct error {
    error() = default;
    error(state s, int32_t c = 0, const std::string& m = std::string("")) : err_state(s), code((uint32_t)c), message(m) {}
    operator state() const { return err_state; }
    state err_state{OK};
    uint32_t code{0};
    std::string message;
};
inline void handle_error(const error& e)
{
    if (e.err_state == WARN) std::fprintf(stderr, "[WARN] %s\n", e.message.c_str());
    if (e.err_state == FATAL) std::fprintf(stderr, "[ERROR] %s\n", e.message.c_str());
}
#endif
} // namespace errors

#define RND_ERR_WHERE(m) (std::string(m) + " in file " + __FILE__ + " at line " + std::to_string(__LINE__))
#ifndef NO_EXCEPTIONS
#define ERROR_TYPE void
#define RAISE_ERROR_WARN(code, m) throw errors::error(errors::WARN, code, RND_ERR_WHERE(m))
#define RAISE_ERROR_FATAL(code, m) throw errors::error(errors::FATAL, code, RND_ERR_WHERE(m))
#define RAISE_ERROR_OK() return
#define PASS_ERROR(code) code
#define HANDLE_ERROR(code)                                                                          \
    try { code; } catch (const errors::error& e) {                                                  \
        std::fprintf(stderr, e.err_state == errors::FATAL ? "[ERROR] %s\n" : "[WARN] %s\n", e.what()); \
    } catch (const std::exception& e) { std::fprintf(stderr, "[ERROR] %s\n", e.what()); }
#else
#define ERROR_TYPE errors::error
#define RAISE_ERROR_WARN(code, m) return errors::error(errors::WARN, code, RND_ERR_WHERE(m))
#define RAISE_ERROR_FATAL(code, m) return errors::error(errors::FATAL, code, RND_ERR_WHERE(m))
#define RAISE_ERROR_OK() return errors::error()
#define PASS_ERROR(code) if (const auto err_ = code; err_ != errors::OK) { return err_; }
#define HANDLE_ERROR(code) errors::handle_error(code)
#endif

// a failed library call is fatal, with the library's own message attached
#define RND_CHECK(call)                                                           \
    do {                                                                          \
        const int rc_ = (call);                                                   \
        if (rc_ != RND_SUCCESS) RAISE_ERROR_FATAL(rc_, std::string(#call ": ") + rnd_last_error()); \
    } while (0)
