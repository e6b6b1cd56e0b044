// Go semantics the translator (tools/go2cpp) must carry, as small functions with results that follow from the Go specification
// (tests/test_go2cpp_semantics.py holds the expected values and says which rule each one exercises). Test infrastructure.
package semantics

import (
	"errors"
	"fmt"
)

const big = 1 << 40 // untyped: fits no 32-bit type, still fine as a constant
const mask = 0xFFFFFFFFFFFFFFFF

type pair struct {
	a, b int
}

type shape interface {
	Area() int
	Name() string
}

type rect struct{ w, h int }
type square struct{ s int }

func (r *rect) Area() int     { return r.w * r.h }
func (r *rect) Name() string  { return "rect" }
func (s square) Area() int    { return s.s * s.s }
func (s square) Name() string { return "square" }

// shifts: a count >= the width gives 0 (or the sign for a signed right shift); unsigned wrap-around; arithmetic >> on negatives
func Shifts(n uint) (uint64, uint32, int32, int64, uint8) {
	one, hi, neg16, neg1, u200 := uint64(1), uint32(0x80000001), int32(-16), int64(-1), uint8(200)
	a := one << n      // n = 64 -> 0
	b := hi << 1       // wraps to 2
	c := neg16 >> 2    // -4
	d := neg1 >> (n + 6) // -1 (count 70)
	e := u200 + 100    // 44
	return a, b, c, d, e
}

// untyped constants take the type of their context; `x := 5` is an int (64 bit); constant expressions are exact
func Consts() (int, uint64, byte, int64, uint32) {
	x := 5
	y := x << 40                 // an int is 64 bits wide
	var m uint64 = mask          // 2^64 - 1
	var b byte = big >> 36       // 16
	z := int64(big) * 1024       // 2^50
	var w uint32 = 1<<32 - 1     // the constant expression is exact
	return y, m, b, z, w
}

// operator precedence: & binds tighter than +, << tighter than + (unlike C), comparison lower than both
func Precedence(a, b uint32) (uint32, uint32, bool, uint32) {
	p := a + b&0xF         // a + (b & 0xF)
	q := a<<2 + b          // (a << 2) + b
	r := a&b == 0          // (a & b) == 0
	s := a &^ b | b>>1     // (a &^ b) | (b >> 1)
	return p, q, r, s
}

// integer division truncates toward zero, % takes the sign of the dividend; unary ^ is bitwise not; conversions truncate / sign-extend
func Arithmetic() (int, int, int, uint8, int8, uint64, int32) {
	m7, p7, two, u8v, i32v, u64v := -7, 7, 2, uint8(200), int32(-1), uint64(1)<<40+7
	a := m7 / two        // -3
	b := m7 % two        // -1
	c := p7 % -two       // 1
	d := ^uint8(0x0F)    // 0xF0
	e := int8(u8v)       // -56
	f := uint64(i32v)    // sign extension: 2^64 - 1
	g := int32(u64v)     // truncation: 7
	return a, b, c, d, e, f, g
}

// slices are views: append within capacity writes through to the shared array, beyond it copies; copy handles overlap; 3-index slices cap
func Slices() (int, int, byte, byte, int, int, byte) {
	base := make([]byte, 4, 8)
	for i := range base {
		base[i] = byte(i + 1)
	}
	view := base[1:3]            // len 2, cap 7
	view = append(view, 99)      // writes base[3]
	x := base[3]                 // 99
	capped := base[0:2:2]
	capped = append(capped, 77)  // must copy: base[2] stays
	y := base[2]                 // 3
	copy(base[1:], base[0:3])    // overlapping: 1 1 2 99 -> base = 1,1,2,3? (memmove semantics: base[1..3] = old base[0..2])
	return len(view), cap(view), x, y, len(capped), cap(base[2:]), base[3]
}

// arrays are values: assignment and range copy them; a pointer to an array ranges without the copy
func Arrays() (int, int, int) {
	a := [4]int{1, 2, 3, 4}
	b := a
	b[0] = 100
	sum := 0
	for i, v := range a { // ranges over a COPY of a: the writes below are not seen
		a[3] = 1000
		sum += v * (i + 1)
	}
	return a[0], b[0], sum // 1, 100, 1+4+9+16 = 30
}

// parallel assignment evaluates the right side first; := in a nested scope shadows; named results and bare return
func Scopes(n int) (x int, y int, err error) {
	x, y = 1, 2
	x, y = y, x+y // 2, 3
	if n > 0 {
		x := x * 10 // shadows
		y = x + 1   // 21
	}
	if n > 5 {
		err = errors.New("big")
		return
	}
	return x, y, nil
}

// switch: no fall-through unless asked; break leaves the switch, not the loop; continue with a label; goto
func Control(limit int) (int, int, int) {
	hits, falls, outer := 0, 0, 0
Outer:
	for i := 0; i < limit; i++ {
		for j := 0; j < limit; j++ {
			switch {
			case j == 1:
				continue Outer
			case i == 3:
				break Outer
			case i == 2:
				falls++
				fallthrough
			default:
				hits++
				if hits > 100 {
					break // leaves the switch only
				}
			}
			outer++
		}
	}
	k := 0
Again:
	k++
	if k < 3 {
		goto Again
	}
	return hits, falls, outer*10 + k
}

// defer runs last in first out at function exit, sees and changes named results; recover stops a panic; a runtime panic is an error
func Deferred(trigger int) (res int, msg string) {
	defer func() {
		res += 1
	}()
	defer func() {
		if r := recover(); r != nil {
			switch v := r.(type) {
			case error:
				msg = "error:" + v.Error()[0:14]
			case string:
				msg = "string:" + v
			default:
				msg = "other"
			}
			res = 100
		}
	}()
	data := []int{1, 2, 3}
	if trigger == 1 {
		panic("boom")
	}
	if trigger == 2 {
		return data[trigger+5], "unreachable" // index out of range: a runtime error
	}
	return data[trigger] * 10, "fine"
}

// interfaces are satisfied structurally, by pointer and by value receivers; closures capture variables, not values
func Interfaces() (int, string, int) {
	shapes := []shape{&rect{w: 3, h: 4}, &square{s: 5}} // (*square's method set holds the value-receiver methods)
	total := 0
	names := ""
	for _, s := range shapes {
		total += s.Area()
		names += s.Name()[0:1]
	}
	counter := 0
	inc := func(by int) int {
		counter += by
		return counter
	}
	inc(2)
	inc(3)
	return total, names, counter
}

// maps are references, a missing key reads as the zero value, comma-ok; strings index as bytes; fmt verbs
func MapsAndStrings() (int, bool, int, byte, string, int) {
	m := make(map[string]int)
	alias := m
	alias["a"] = 7
	v, ok := m["b"]
	s := "héllo" // é is two bytes in UTF-8
	p := pair{a: 1, b: 2}
	q := p
	q.a = 50
	missing := m["nothing"] // a read of a missing key: the zero value, and no entry appears
	return m["a"] + v + missing, ok, len(s), s[1], fmt.Sprintf("%d-%s", p.a+q.a, "x"), len(m)
}

// x.(T) on an element of a map of interfaces is an assertion on the element (kanzi-go: ctx["from"].(int), io/CompressedStream.go:1150): an int is an
// int and not a uint, a missing key is a nil interface and asserts to nothing
func MapAsserts() (int, bool, bool, bool) {
	ctx := make(map[string]any)
	ctx["from"] = 3
	ctx["jobs"] = uint(4)
	a, okA := ctx["from"].(int)
	_, okB := ctx["jobs"].(int)
	_, okC := ctx["to"].(int)
	v := ctx["from"]
	b, _ := v.(int)
	return a + b, okA, okB, okC
}

type counter struct {
	n    int
	hist [3]int
}

func (c counter) bumpCopy() int { // value receiver: works on a copy
	c.n += 10
	c.hist[0] = 99
	return c.n
}

func (c *counter) bump() int {
	c.n += 10
	c.hist[0]++
	return c.n
}

// structs (and the arrays inside them) are values; value receivers get a copy; swaps through parallel assignment; append with a spread;
// an untyped constant shifted by a variable takes its type from the context
func Values(n uint) (int, int, int, int, int, int, uint32, int) {
	c := counter{n: 1}
	d := c
	d.hist[1] = 7
	r1 := c.bumpCopy() // 11, c unchanged
	r2 := c.bump()     // 11, c.n = 11, c.hist[0] = 1
	a := []int{1, 2, 3, 4}
	a[0], a[3] = a[3], a[0]
	b := append([]int{9}, a[1:3]...)
	var m uint32 = 1<<n - 1 // n = 40: the shift happens in uint32 -> 0, minus 1 wraps
	k := 1 << n             // an int: 2^40
	return r1, r2, c.n + c.hist[0]*100 + c.hist[1], d.hist[1], a[0]*10 + a[3], len(b)*100 + b[2], m, k
}

type tape struct {
	log  int
	data []int
}

func (t *tape) next(v int) int { // leaves a trace of the order it was called in
	t.log = t.log*10 + v
	return v
}

func (t *tape) slot(v int) *int {
	t.log = t.log*10 + v
	return &t.data[v]
}

func three(a, b, c int) int { return a*100 + b*10 + c }

// Go runs the calls inside one expression in lexical left-to-right order, whatever the operator or the position: operands of | and +, arguments,
// several returned values, slice bounds, an index on the left of an assignment before the right side. C++ leaves most of these open.
func CallOrder() (int, int, int, int, int, int, int) {
	t := &tape{data: make([]int, 8)}
	a := t.next(1)<<2 | t.next(2)<<1 | t.next(3) // 1,2,3
	l1 := t.log
	t.log = 0
	b := three(t.next(4), t.next(5), t.next(6)) + t.next(7) // 4,5,6,7
	l2 := t.log
	t.log = 0
	t.data[t.next(1)] = t.next(2) // index first, then the value: 1,2
	*t.slot(3) += t.next(4)       // 3,4
	s := t.data[t.next(0):t.next(5)]
	l3 := t.log // 123405
	t.log = 0
	x, y := pairOf(t)
	return a, l1, b, l2, l3*10 + len(s), x*10 + y, t.log
}

func pairOf(t *tape) (int, int) {
	return t.next(8), t.next(9) // 8,9
}

// what table-driven tests lean on: struct types declared inside a function, anonymous struct types, map literals, a range variable captured by a
// closure that outlives its iteration (Go >= 1.22: every iteration has its own copy), x.(T) on a non-empty interface, comma-ok included
func TableDriven() (int, int, int, int, bool, string) {
	type row struct {
		name string
		w, h int
	}
	rows := []row{{name: "a", w: 2, h: 3}, {"b", 4, 5}}
	anon := []struct {
		k string
		v int
	}{{"x", 1}, {"y", 2}}
	m := map[string]int{"one": 1, "two": 2}
	var fs []func() int

	for _, r := range rows {
		fs = append(fs, func() int { return r.w * r.h })
	}

	sum := 0

	for _, f := range fs {
		sum = sum*100 + f() // 6 then 20: each closure kept ITS r
	}

	var s shape = &rect{w: 3, h: 4}
	rc := s.(*rect)
	_, isSquare := s.(*square)
	names := ""

	for _, a := range anon {
		names += a.k
	}

	return sum, anon[0].v + anon[1].v*10, m["one"] + m["two"]*10 + len(m)*100, rc.w*10 + rc.h, isSquare, names + rows[1].name
}

// fmt verbs with flags, width and precision as the reference's messages use them (%.8b for skip flags, %x for checksums)
func Formats() string {
	flags := byte(31)
	sum := uint64(0xabcdef)
	return fmt.Sprintf("[%.8b] [%x] [%X] [%5d] [%-5d] [%05d] [%q] [%v] [%d%%]", flags, sum, uint32(255), -42, 7, -42, "hi", 3, 50)
}
