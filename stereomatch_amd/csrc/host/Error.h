// Error.h -- the host driver's exception type.  Same role and name as the reference's CError
// (Error.h:23-32): a printf-formatted message thrown up to main(), which prints it and
// returns -1 (main.cpp:44-49).  Non-zero smhip status codes are converted into it.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <exception>

struct CError : public std::exception {
    char message[1024];
    explicit CError(const char *fmt, ...)
    {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(message, sizeof(message), fmt, ap);
        va_end(ap);
    }
    const char *what() const noexcept override { return message; }
};
