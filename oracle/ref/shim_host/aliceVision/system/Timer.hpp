// oracle/_ref (host): stand-in for aliceVision/system/Timer.hpp (only named by the code under test)
#pragma once
namespace aliceVision { namespace system {
class Timer
{
  public:
    double elapsedMs() const { return 0.0; }
    double elapsed() const { return 0.0; }
};
}} // namespace aliceVision::system
