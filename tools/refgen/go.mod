module knzrefgen

go 1.24

require github.com/flanglet/kanzi-go/v2 v2.0.0

// tools/make_ref_vectors.sh points this at the kanzi-go checkout it is given (go mod edit -replace ...)
replace github.com/flanglet/kanzi-go/v2 => ../../../reference/v2
