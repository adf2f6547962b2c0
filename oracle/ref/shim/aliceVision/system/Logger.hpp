/* STAND-IN for aliceVision/system/Logger.hpp (Boost.Log in the reference): oracle/_ref test infrastructure only */
#pragma once
#include <iostream>
#include <sstream>
#include <stdexcept>
#define ALICEVISION_LOG_TRACE(a) do { } while(0)
#define ALICEVISION_LOG_DEBUG(a) do { } while(0)
#define ALICEVISION_LOG_INFO(a) do { } while(0)
#define ALICEVISION_LOG_WARNING(a) do { std::cerr << "[ref] warning: " << a << std::endl; } while(0)
#define ALICEVISION_LOG_ERROR(a) do { std::cerr << "[ref] error: " << a << std::endl; } while(0)
#define ALICEVISION_THROW(EXCEPTION, x) { std::stringstream s; s << x; throw EXCEPTION(s.str()); }
#define ALICEVISION_THROW_ERROR(x) ALICEVISION_THROW(std::runtime_error, x)
