#!/bin/bash
# Re-cut tools/experiments/lab_scaffolding.patch against the current product sources: apply it (with fuzz) to a copy of
# ndzip_amd/csrc, stop on rejects (fix them in the copy by hand: the path is printed), otherwise write the fresh diff back.
# usage: tools/experiments/refresh_lab_patch.sh [dir-of-an-already-fixed-copy]
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
work=${1:-$(mktemp -d)}
if [ -z "$1" ]; then
  cp "$root"/ndzip_amd/csrc/*.hip "$root"/ndzip_amd/csrc/*.hpp "$root"/ndzip_amd/csrc/*.inl "$work"/
  if ! patch -s -p1 -F3 -d "$work" < "$root/tools/experiments/lab_scaffolding.patch"; then
    echo "rejects in $work: fix them there, then run: $0 $work" >&2; exit 1
  fi
fi
rm -f "$work"/*.rej "$work"/*.orig
tmp=$(mktemp -d); mkdir "$tmp/a" "$tmp/b"
cp "$root"/ndzip_amd/csrc/*.hip "$root"/ndzip_amd/csrc/*.hpp "$root"/ndzip_amd/csrc/*.inl "$tmp/a/"
cp "$work"/*.hip "$work"/*.hpp "$work"/*.inl "$tmp/b/"
(cd "$tmp" && diff -u a b | sed -E 's#^(---|\+\+\+) ([ab]/[^\t]*)\t.*#\1 \2#' | grep -v "^diff -u\|^Only in") > "$root/tools/experiments/lab_scaffolding.patch" || true
wc -l "$root/tools/experiments/lab_scaffolding.patch"
