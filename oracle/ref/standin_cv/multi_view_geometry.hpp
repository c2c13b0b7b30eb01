// shadows /root/reference/include/multi_view_geometry.hpp for the build of feature_tracker.cpp: that file includes it and uses nothing of it
#pragma once
