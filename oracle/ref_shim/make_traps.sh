#!/bin/sh
# ORACLE — TEST INFRASTRUCTURE ONLY.
# usage: make_traps.sh <half-linked .so> > traps.s
# Emits one trapping definition for every poselib:: symbol the library still lacks, i.e. the entry points of the
# reference translation units Makefile.ref leaves out (solver families that are not on the LO-RANSAC path).  The
# library then loads under RTLD_NOW; calling any of them executes ud2.
echo '	.text'
nm -D --undefined-only "$1" | awk '{print $NF}' | grep -E '^_ZNK?7poselib' | sort -u | while read -r sym; do
    printf '\t.globl %s\n\t.type %s,@function\n%s:\n\tud2\n' "$sym" "$sym" "$sym"
done
echo '	.section .note.GNU-stack,"",@progbits'
