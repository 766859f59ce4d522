#!/bin/bash
P=tools/coresidency_probe
echo "mode 1: 6 in-range pieces per wave at the start";            timeout 300 $P 300 0 17 0 6 0 | tail -1
echo "mode 3: 6 in-range + 6 out-of-range";                        timeout 300 $P 300 0 19 0 6 6 | tail -1
echo "mode 2: 6 out-of-range only";                                timeout 300 $P 300 0 18 0 0 6 | tail -1
echo "mode 5: pause, then 6 in-range";                             timeout 300 $P 300 0 21 0 6 0 | tail -1
echo "mode 7: pause, 6 in-range + 6 out-of-range";                 timeout 300 $P 300 0 23 0 6 6 | tail -1
echo "mode 11: 6 in-range + 6 out-of-range, pause before the end"; timeout 300 $P 300 0 27 0 6 6 | tail -1
echo "mode 1: 18 in-range pieces per wave";                        timeout 300 $P 300 0 17 0 18 0 | tail -1
echo "mode 0: no DMA at all (LDS allocated)";                      timeout 300 $P 300 0 16 0 0 0 | tail -1
