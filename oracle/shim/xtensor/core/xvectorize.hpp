// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/core/xvectorize.hpp)
#pragma once
#include "../../xt_shim.hpp"
