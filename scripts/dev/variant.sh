#!/bin/bash
# dev helper: build a variant of the library in bisect/<name>/ after applying a sed script / patch command to its csrc copy.
# usage: scripts/dev/variant.sh <name> '<shell command run inside bisect/<name>/poasta_amd/csrc>'
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
d=$root/bisect/$name
rm -rf "$d"; mkdir -p "$d/poasta_amd" "$d/scripts"
cp -r "$root/include" "$d/"
cp "$root"/poasta_amd/*.py "$d/poasta_amd/"
cp -r "$root/poasta_amd/csrc" "$root/poasta_amd/host" "$d/poasta_amd/"
cp "$root"/scripts/*.py "$d/scripts/"; ln -s ../../oracle "$d/oracle"; ln -s ../../tests "$d/tests"
cd "$d/poasta_amd/csrc"
bash -c "$*"
make ../libpoasta_amd.so > "$d/build.log" 2>&1 || { tail -n 20 "$d/build.log"; exit 1; }
ls -la "$d/poasta_amd/libpoasta_amd.so"
