// container shim header: see ../../xt_shim.hpp (stand-in for xtensor/containers/xadapt.hpp)
#pragma once
#include "../../xt_shim.hpp"
