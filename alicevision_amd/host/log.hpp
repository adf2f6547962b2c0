// log.hpp — severity-filtered logging to stdout in the reference's line format "[HH:MM:SS.uuuuuu][level] message"
// (system/Logger.hpp; levels fatal, error, warning, info, debug, trace selected by --verboseLevel).
#pragma once

#include <chrono>
#include <cstdio>
#include <ctime>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>

namespace avdm_host {

enum class LogLevel { Fatal = 0, Error, Warning, Info, Debug, Trace };

struct Logger
{
    static LogLevel& level()
    {
        static LogLevel l = LogLevel::Info;
        return l;
    }
    static bool setLevel(const std::string& s)
    {
        if(s == "fatal") level() = LogLevel::Fatal;
        else if(s == "error") level() = LogLevel::Error;
        else if(s == "warning") level() = LogLevel::Warning;
        else if(s == "info") level() = LogLevel::Info;
        else if(s == "debug") level() = LogLevel::Debug;
        else if(s == "trace") level() = LogLevel::Trace;
        else return false;
        return true;
    }
    static void write(LogLevel l, const char* name, const std::string& msg)
    {
        if((int)l > (int)level())
            return;
        static std::mutex m;
        std::lock_guard<std::mutex> lock(m);
        const auto now = std::chrono::system_clock::now();
        const std::time_t t = std::chrono::system_clock::to_time_t(now);
        const long us = (long)(std::chrono::duration_cast<std::chrono::microseconds>(now.time_since_epoch()).count() % 1000000);
        std::tm tmv;
        localtime_r(&t, &tmv);
        char buf[32];
        std::snprintf(buf, sizeof(buf), "[%02d:%02d:%02d.%06ld]", tmv.tm_hour, tmv.tm_min, tmv.tm_sec, us);
        std::cout << buf << "[" << name << "] " << msg << std::endl;
    }
};

#define AVDM_LOG_AT(lvl, name, expr)                                                                                                                          \
    do                                                                                                                                                        \
    {                                                                                                                                                         \
        if((int)(lvl) <= (int)::avdm_host::Logger::level())                                                                                                   \
        {                                                                                                                                                     \
            std::ostringstream avdm_log_os;                                                                                                                   \
            avdm_log_os << expr;                                                                                                                              \
            ::avdm_host::Logger::write(lvl, name, avdm_log_os.str());                                                                                         \
        }                                                                                                                                                     \
    } while(0)
#define AVDM_LOG_TRACE(expr) AVDM_LOG_AT(::avdm_host::LogLevel::Trace, "trace", expr)
#define AVDM_LOG_DEBUG(expr) AVDM_LOG_AT(::avdm_host::LogLevel::Debug, "debug", expr)
#define AVDM_LOG_INFO(expr) AVDM_LOG_AT(::avdm_host::LogLevel::Info, "info", expr)
#define AVDM_LOG_WARNING(expr) AVDM_LOG_AT(::avdm_host::LogLevel::Warning, "warning", expr)
#define AVDM_LOG_ERROR(expr) AVDM_LOG_AT(::avdm_host::LogLevel::Error, "error", expr)
#define AVDM_THROW_ERROR(expr)                                                                                                                                \
    do                                                                                                                                                        \
    {                                                                                                                                                         \
        std::ostringstream avdm_err_os;                                                                                                                       \
        avdm_err_os << expr;                                                                                                                                  \
        throw std::runtime_error(avdm_err_os.str());                                                                                                          \
    } while(0)

} // namespace avdm_host
