# dev: A/B of two builds of the Gram kernel: lib/ against lib_oldgram/ (built by hand from an older csrc/gram.hip).  GPU box only.
cd $GRAFT_REPO_ROOT
P=bayesian-coresets_amd
S="400,512 999,512 1497,1024 1024,1024 2048,2048 512,4096 1497,2048"
for r in 1 2; do
echo new; python tools/gram_bench.py $S 2>&1 | grep "k=" | cut -c1-45
mv $P/lib $P/lib_new && mv $P/lib_oldgram $P/lib
echo old; python tools/gram_bench.py $S 2>&1 | grep "k=" | cut -c1-45
mv $P/lib $P/lib_oldgram && mv $P/lib_new $P/lib
done
