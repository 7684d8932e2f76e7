// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/containers/xtensor.hpp)
#pragma once
#include "../../xt_shim.hpp"
