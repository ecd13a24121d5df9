tools/abn.sh 2 "--pmc off --tf tf1" ab_libs/cur.so ab_libs/alpha6.so
tools/abn.sh 2 "--pmc off --tf tf1 --scene ct" ab_libs/cur.so ab_libs/alpha6.so
tools/abn.sh 2 "--pmc off --hints off" ab_libs/cur.so ab_libs/alpha6.so
tools/abn.sh 2 "--pmc off --hints off --vol 256 --img 256 --grads none --steps 20" ab_libs/cur.so ab_libs/alpha6.so
tools/abn.sh 2 "--pmc off --tf tf1 --cam inside --steps 3" ab_libs/cur.so ab_libs/alpha6.so
