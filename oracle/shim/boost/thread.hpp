// TEST INFRASTRUCTURE ONLY: see ptor_boost_shim.hpp
#include "ptor_boost_shim.hpp"
