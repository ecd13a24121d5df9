tools/abn.sh 2 "--pmc off" ab_libs/acfg.so ab_libs/dmask.so
tools/abn.sh 2 "--pmc off --tf tf1" ab_libs/acfg.so ab_libs/dmask.so
tools/abn.sh 2 "--pmc off --tf tf1 --scene ct" ab_libs/acfg.so ab_libs/dmask.so
tools/abn_opt.sh 2 ab_libs/acfg.so ab_libs/dmask.so
