#!/bin/bash
# The worker pool of the host side under ThreadSanitizer (no GPU needed: a host-only context).  Builds the library's objects with
# -fsanitize=thread on the host side into merian-quake_amd/build/tsan, links tools/tsan_pool_driver.cpp against them and runs it.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/merian-quake_amd B=build/tsan OUT=build/tsan/libmqhip_tsan_unused.so EXTRA="-Xarch_host -fsanitize=thread -g" LDEXTRA="-fsanitize=thread" > /dev/null 2>&1 || true
g++ -O1 -g -fsanitize=thread -I$root/include -c $root/tools/tsan_pool_driver.cpp -o $root/merian-quake_amd/build/tsan/driver.o
/opt/rocm/bin/hipcc -fsanitize=thread $root/merian-quake_amd/build/tsan/driver.o $root/merian-quake_amd/build/tsan/mq_*.o -o $root/merian-quake_amd/build/tsan/driver --offload-arch=gfx950 2> /dev/null
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" $root/merian-quake_amd/build/tsan/driver > $root/merian-quake_amd/build/tsan/out.txt 2>&1 || true
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $root/merian-quake_amd/build/tsan/out.txt)"
tail -1 $root/merian-quake_amd/build/tsan/out.txt | cut -c1-200
