#include "../../../oracle/ref/standin/sophus/so3.hpp"
