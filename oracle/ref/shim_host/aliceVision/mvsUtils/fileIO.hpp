// oracle/_ref (host): STAND-IN for aliceVision/mvsUtils/fileIO.hpp — the one function the code under test calls, for the names of files it
// would export (never written here: the export switches stay off) and as the key of the in-memory map store.  Test infrastructure only.
#pragma once
#include <string>

#include <aliceVision/image/Image.hpp>
#include <aliceVision/mvsUtils/MultiViewParams.hpp>

namespace aliceVision {
namespace mvsUtils {
inline std::string getFileNameFromIndex(const MultiViewParams&, int index, EFileType fileType, const std::string& customSuffix = "", int tileBeginX = -1,
                                        int tileBeginY = -1)
{
    return std::to_string(index) + ":" + std::to_string((int)fileType) + customSuffix + ":" + std::to_string(tileBeginX) + ":" + std::to_string(tileBeginY);
}
} // namespace mvsUtils
} // namespace aliceVision
