// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/io/xio.hpp)
#pragma once
#include "../../xt_shim.hpp"
