// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/core/xtensor_forward.hpp)
#pragma once
#include "../../xt_shim.hpp"
