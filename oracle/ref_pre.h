/* TEST INFRASTRUCTURE ONLY (oracle/).  Force-included (-include) when the reference's
 * encoder/EncoderContext.cpp is compiled in place from /root/reference.
 *
 * That TU includes "dirent.h" (EncoderContext.cpp:8899), which resolves to the reference's bundled
 * Win32 port of dirent (encoder/dirent.h:28 pulls <windows.h>).  On this POSIX image the real
 * <dirent.h> is the system's own header, so we include it first and set the port's include guard
 * (encoder/dirent.h:9-10) so the port is skipped.  Nothing is emulated: the directory-walking code
 * (LUT bank loading, out of scope) simply binds to glibc's opendir/readdir.
 */
#include <dirent.h>
#define DIRENT_H
