# builds several body variants in parallel: tools/build_many.sh "name:opts" "name:opts" ...   (4 at a time)
root=$(cd "$(dirname "$0")/.." && pwd)
printf '%s\n' "$@" | xargs -P 4 -I{} bash -c 'v="{}"; n=${v%%:*}; o=${v#*:}; bash '$root'/tools/build_variant.sh $n $o 2>&1 | tail -1'
