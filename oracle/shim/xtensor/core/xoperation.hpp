// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/core/xoperation.hpp)
#pragma once
#include "../../xt_shim.hpp"
