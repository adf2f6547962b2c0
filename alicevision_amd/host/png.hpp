// png.hpp — the one PNG flavour the depth-map filtering step exchanges between its two passes: 8-bit greyscale, non-interlaced
// (the modal-count map `<viewId>_nmodMap.png`, fuseCut/Fuser.cpp:220-223, mvsUtils/fileIO.cpp:291-297).  zlib only.
#pragma once

#include <string>
#include <vector>

namespace avdm_host {

void writePngGray8(const std::string& path, int width, int height, const unsigned char* data);
// 8-bit greyscale (colour type 0) or 8-bit RGB / RGBA / grey+alpha (first channel is returned), non-interlaced; throws otherwise
void readPngGray8(const std::string& path, int& width, int& height, std::vector<unsigned char>& data);

} // namespace avdm_host
