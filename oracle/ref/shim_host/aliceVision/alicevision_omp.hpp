// oracle/_ref (host): stand-in for aliceVision/alicevision_omp.hpp
#pragma once
#include <omp.h>
