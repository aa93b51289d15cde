#!/bin/bash
# ONE command that turns every [Lucene-recall] item of SURVEY.md 8(c) into a committed golden, on a box with a JDK (>= 21) and
# lucene-core 10.x (the reference pins 10.4.0: gradle/libs.versions.toml:7) -- neither exists in the build image, so this has
# never run there:
#     LUCENE_JARS=/path/to/lucene-core-10.4.0.jar bash scripts/make_lucene_goldens.sh
# writes tests/golden/lucene_shapes.json (bench/lucene/LuceneGolden.java over the 20 k-doc fixture of scripts/dump_corpus.py:
# plain disjunctions -- multi-term (float)sum(double) --, DisjunctionMax with tie breakers 0 / 0.3 / 1, MUST next to SHOULD
# clauses, minimumNumberShouldMatch 2 / 3, a FILTER and a MUST_NOT doc set, a BoostQuery; doc lengths far above 40 tokens; both
# totalHits relations) and, with BASELINE=1, the C2 timing + parity run of bench/lucene/LuceneBaseline.java.
# Then: python -m pytest tests/test_lucene_golden.py   (the oracle on any box, the device with -m gpu)
set -eu
cd "$(dirname "$0")/.."; ROOT=$(pwd)
: "${LUCENE_JARS:?set LUCENE_JARS to the lucene-core jar (classpath)}"
command -v javac >/dev/null && command -v java >/dev/null || { echo "no JDK on this box"; exit 2; }
T=$(mktemp -d)
python scripts/dump_corpus.py --fixture "$T/dump"
javac -cp "$LUCENE_JARS" bench/lucene/LuceneBaseline.java bench/lucene/LuceneGolden.java -d "$T/cls"
java -cp "$T/cls:$LUCENE_JARS" LuceneGolden "$T/dump" "$T/index" "$T/lucene_shapes.json"
cp "$T/lucene_shapes.json" tests/golden/lucene_shapes.json
echo "wrote tests/golden/lucene_shapes.json: $(python -c "import json; d=json.load(open('tests/golden/lucene_shapes.json')); print(len(d['queries']), 'queries, lucene', d['lucene'], 'java', d['java'])")"
if [ "${BASELINE:-0}" = "1" ]; then
  python scripts/dump_corpus.py C2 "$T/c2" 1024
  java -Xmx16g -cp "$T/cls:$LUCENE_JARS" LuceneBaseline "$T/c2" "$T/c2index" "$(nproc)" "$T/lucene_c2.json" 1024
  python - "$T/lucene_c2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("LuceneBaseline C2:", {k: d[k] for k in ("lucene", "java", "threads", "queries", "queries_per_s", "p50_ms", "p99_ms")})
PY
fi
rm -rf "$T"
