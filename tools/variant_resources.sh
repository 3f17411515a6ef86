#!/bin/bash
# kernel resource summary of a variant built by tools/build_variant.sh:  tools/variant_resources.sh <name> [grep pattern]
python $(dirname $0)/kernel_resources.py /tmp/atl_variant_$1/resource.txt | grep -E "${2:-.}"
