#include "so3.hpp"
