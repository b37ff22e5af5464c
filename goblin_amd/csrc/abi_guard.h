// No C++ exception crosses the C ABI: every extern "C" entry point that can allocate or parse
// runs its body through gbl_guard, which turns std::bad_alloc into GBL_ERR_OOM and anything else into GBL_ERR_INTERNAL
// with the exception's text as the error message.
#pragma once
#include <exception>
#include <new>
#include <string>

#include "../../include/goblin_hip.h"

template <class Body, class OnError>
inline gbl_status gbl_guard(Body&& body, OnError&& on_error) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { on_error(std::string("out of host memory")); } catch (...) {}
        return GBL_ERR_OOM;
    } catch (const std::exception& e) {
        try { on_error(std::string("internal error: ") + e.what()); } catch (...) {}
        return GBL_ERR_INTERNAL;
    } catch (...) {
        try { on_error(std::string("internal error: unknown exception")); } catch (...) {}
        return GBL_ERR_INTERNAL;
    }
}
