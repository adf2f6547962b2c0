// oracle/_ref (host): see accumulators.hpp
#pragma once
#include <boost/accumulators/accumulators.hpp>
