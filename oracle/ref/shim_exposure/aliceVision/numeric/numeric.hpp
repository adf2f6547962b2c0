// stand-in for aliceVision/numeric/numeric.hpp (Eigen typedefs and helpers the exposure code does not use): test infrastructure
#pragma once
#include <cmath>
#include <algorithm>
#include <iostream>
