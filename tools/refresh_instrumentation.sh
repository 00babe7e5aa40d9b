#!/bin/bash
# Re-bases tools/experiments/frames_instrumentation.patch on the current csrc/: applies it to a copy (rejects allowed), lets you fix the
# rejected hunks by hand in /tmp/sp_instr/new, and writes the patch again.
#   tools/refresh_instrumentation.sh apply     -> /tmp/sp_instr/{old,new}; lists *.rej
#   tools/refresh_instrumentation.sh write     -> regenerates the patch from /tmp/sp_instr/{old,new}
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
W=/tmp/sp_instr
case "$1" in
apply)
  rm -rf $W && mkdir -p $W/old/spectroplot-js_amd $W/new/spectroplot-js_amd
  for d in old new; do cp -r $REPO/spectroplot-js_amd/csrc $W/$d/spectroplot-js_amd/; cp -r $REPO/include $W/$d/; done
  (cd $W/new && patch -p1 < $REPO/tools/experiments/frames_instrumentation.patch || true)
  find $W/new -name "*.rej" ;;
write)
  find $W/new -name "*.rej" -o -name "*.orig" | xargs rm -f
  (cd $W && diff -ruN old/spectroplot-js_amd new/spectroplot-js_amd || true) | sed 's#^--- old/#--- a/#; s#^+++ new/#+++ b/#; s#^diff -ruN old/\(\S*\) new/\(\S*\)#diff -ruN a/\1 b/\2#' > $REPO/tools/experiments/frames_instrumentation.patch
  wc -l $REPO/tools/experiments/frames_instrumentation.patch ;;
*) echo "usage: $0 apply|write"; exit 2 ;;
esac
